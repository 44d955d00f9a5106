bash tools/collect.sh abtrain MCQUIC_AMD_PAIR_RULE 0 1 2
for r in 1 2; do for v in 0 1; do MCQUIC_AMD_PAIR_RULE=$v python bench.py --batch 1 --graphs --steps 50 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch1 PAIR_RULE=$v', d['ms_per_step'])"; done; done
MCQUIC_AMD_PAIR_RULE=1 timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
