run() { echo -n "$1 :: "; shift; python tools/bench_train.py --graph --steps 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
run base
for t in 0x41 0x141 0x121 0x11 0x21 0x122 0x22; do run x1_64_$t --tile-rule 1:64:128:$t; done
run base_again
for t in 0x121 0x11 0x21 0x141; do run x1_32_$t --tile-rule 1:32:128:$t; done
run base_3
