cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_entropy_coder.py -q -m gpu 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06o/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['secondary']['speed_protocol'])[:400])
print(json.dumps(d['secondary']['two_core_host'])[:600])
PY
for f in 1 0 1 0; do echo "overlap=$f: $(MCQUIC_AMD_CODER_OVERLAP=$f python tools/bench_speed_protocol.py 2>/dev/null | tail -1)"; done
