# after a change to vq_logits_kernel: parity first, then the kernel alone, then the captured training step
timeout 900 python -m pytest tests/test_gpu_train_forward.py tests/test_gpu_backward.py -x -q 2>&1 | tail -3
python tools/probes/time_logits.py 2>&1 | tail -8
for i in 1 2 3; do python tools/bench_train.py --steps 20 --graph 2>/dev/null | tail -1 | cut -c100-200; done
