run() { echo -n "$1 :: "; shift; python tools/bench_train.py --graph --steps 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['grad_norm'])"; }
run base_nor16
run r16
run r16_again
run r16_4096
run R1_x1_121 --tile-rule 1:128:128:0x121,1:64:128:0x121,1:32:128:0x121
run R2_x1_11 --tile-rule 1:128:128:0x11,1:64:128:0x11
run R3_x1_128_22 --tile-rule 1:128:128:0x22
run R3b_x1_128_21 --tile-rule 1:128:128:0x21,1:64:128:0x21
run R4_x24_32 --tile-rule 2:32:128:0x11,4:32:128:0x11
run R5_shuf_22 --tile-rule 1:64:512:0x22,1:32:512:0x22
run R5_shuf_121 --tile-rule 1:64:512:0x121,1:32:512:0x121
run R5_shuf_11 --tile-rule 1:64:512:0x11,1:32:512:0x11
run R6_x2_64_22 --tile-rule 2:64:128:0x22
run R6_x2_64_121 --tile-rule 2:64:128:0x121
