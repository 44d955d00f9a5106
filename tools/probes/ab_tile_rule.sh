# in-step A/B of forced tiles on the captured training step: alternating runs on one box
# usage: bash tools/probes/ab_tile_rule.sh "<rule A>" "<rule B>" ... (an empty string = the launcher's own rule)
for r in 1 2; do for rule in "$@"; do
  if [ -z "$rule" ]; then a=""; else a="--tile-rule $rule"; fi
  python tools/bench_train.py --steps 20 --graph $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$rule]', d['ms_per_step'], d['loss'])"
done; done
