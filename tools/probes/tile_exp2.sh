run() { echo -n "$1 :: "; shift; python tools/bench_train.py --graph --steps 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['grad_norm'])"; }
run base
run x1_128_41 --tile-rule 1:128:128:0x41
run x2_64_41 --tile-rule 2:64:128:0x41
run x1_64_41 --tile-rule 1:64:128:0x41
run all_41 --tile-rule 1:128:128:0x41,2:64:128:0x41,1:64:128:0x41
run shuf64_41 --tile-rule 1:64:512:0x41
run base_again
