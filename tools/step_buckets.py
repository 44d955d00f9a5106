#!/usr/bin/env python3
"""Buckets of a captured training step's replay from its ordered kernel sequence (tools/probes/dump_step_sequence.py output):
    python tools/step_buckets.py profiles/r04_train_step_sequence.txt profiles/r05_train_step_sequence.txt
launches and ms per bucket -- the table of DESIGN section 4."""
import collections
import sys

ORDER = ["3x3 conv, maps >= 32x32", "3x3 conv, maps <= 16x16", "1x1 conv", "wgrad row walk", "wgrad reduce passes", "wgrad stride 2", "wgrad <= 512 pixels",
         "wgrad 1x1", "soft assignment", "glue: ours", "glue: ATen / runtime"]


def bucket(name: str) -> str:
    n = name
    if "at::native" in n or "rocclr" in n or "elementwise_kernel_with_index" in n:
        return "glue: ATen / runtime"
    # (", 8, 16, 4," = the four-tap walk of a stride-2 layer's input gradient, round 5: a 3x3 launch like the nine-tap one it replaces)
    if "conv_mfma_kernel<4, 2, 0, 9" in n or "conv_mfma_kernel<4, 1, 0, 9" in n or "conv_head16" in n or \
            "conv_mfma_kernel<4, 2, 0, 8, 16, 4," in n or "conv_mfma_kernel<4, 1, 0, 8, 16, 4," in n:
        return "3x3 conv, maps >= 32x32"
    if "conv_mfma_kernel<1, 1, 0, 9" in n or "conv_t16" in n or "conv_mfma_kernel<2, 1, 0, 9" in n or "conv_mfma_kernel<2, 2, 0, 9" in n or \
            ", 0, 8, 16, 4," in n:
        return "3x3 conv, maps <= 16x16"
    if "conv_mfma_kernel" in n:
        return "1x1 conv"
    if "wgrad_rows_kernel" in n:
        return "wgrad row walk"
    if "wgrad_rows_reduce" in n or "wgrad_reduce_kernel" in n:      # (incl. wgrad_rows_reduce_batch_kernel, round 5)
        return "wgrad reduce passes"
    if "wgrad_rows_s2" in n:
        return "wgrad stride 2"
    if "wgrad_rows1" in n or "wgrad_t16_kernel<1" in n:
        return "wgrad 1x1"
    if "wgrad_t16" in n or "conv_wgrad_kernel" in n:
        return "wgrad <= 512 pixels"
    if n.startswith("vq_logits") or "gumbel" in n or "softmax_bwd" in n or "vq_dx" in n or "vq_dc" in n:
        return "soft assignment"
    return "glue: ours"


def main():
    for path in sys.argv[1:]:
        b = collections.defaultdict(lambda: [0, 0.0])
        for ln in open(path):
            if ln.startswith("#"):
                continue
            f = ln.split()
            k = bucket(" ".join(f[4:]))
            b[k][0] += 1
            b[k][1] += float(f[2])
        print(path)
        for k in ORDER:
            print(f"  {k:28s} {b[k][0]:4d} launches {b[k][1] / 1e3:7.2f} ms")
        print(f"  {'total':28s} {sum(v[0] for v in b.values()):4d} launches {sum(v[1] for v in b.values()) / 1e3:7.2f} ms")


if __name__ == "__main__":
    main()
