#!/usr/bin/env python3
"""Where a tiny-map conv launch spends its ~10 us (round-3 probe).  Needs a scratch build of conv_mfma.hip with s_memrealtime
stamps (not in the tree: the STAMP() lines are patched into a copy under mcquic_amd/variants/), loaded
through MCQUIC_AMD_LIB.  The stamp buffer travels in the otherwise unused gate_id pointer.

    MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/stamps.so python tools/probes/tiny_stamps.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcquic_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["entry", "geometry done", "rings issued", "k-loop done", "reduced", "epilogue done"]
for (n, hw, nprob, ks) in ((8, 4, 2, 3), (8, 8, 4, 3), (8, 8, 2, 3), (8, 16, 2, 3), (8, 4, 1, 1)):
    xs = [torch.randn((n, 128, hw, hw), device=dev) for _ in range(nprob)]
    packs = [[ops.PackedConv(torch.randn((128, 128, ks, ks), device=dev) * 0.03, torch.randn(128, device=dev)) for _ in range(nprob)] for _ in range(6)]
    res = [torch.randn((n, 128, hw, hw), device=dev) for _ in range(nprob)]
    big = [torch.zeros((n, 128, hw, hw), device=dev) for _ in range(nprob)]      # gate_id-shaped stamp buffers (problem 0's is used)
    rows = []
    for it in range(12):
        for b in big:
            b.zero_()
        # evict: touch 512 MB so that weights / inputs come from HBM like in a training step
        junk = torch.empty(128 << 20, device=dev).normal_() if it >= 6 else None
        torch.cuda.synchronize()
        pp = [dict(res=r, gate_id=b) for r, b in zip(res, big)]
        if nprob == 1:
            ops.conv2d(xs[0], packs[it % 6][0], 1, dual_silu=True, **pp[0])
        else:
            ops.conv2d_multi(xs, packs[it % 6], 1, per_problem=pp, dual_silu=True)
        torch.cuda.synchronize()
        st = big[0].view(torch.int64).cpu().numpy().reshape(-1)
        nwg = (st.reshape(-1, 16)[:, 0] != 0).sum()
        st = st[: nwg * 16].reshape(nwg, 2, 8)[:, :, :6].astype(np.float64) * 0.01       # us (100 MHz)
        t0 = st[:, 0, 0].min()
        rows.append((it, nwg, st - t0))
    for label, sel in (("warm (MALL)", rows[2:6]), ("cold (after a 512 MB sweep)", rows[7:])):
        w0 = np.stack([r[2][:, 0, :] for r in sel])        # [iter, wg, stamp] wave 0 (the owner)
        w7 = np.stack([r[2][:, 1, :] for r in sel])
        print(f"n{n} {hw}x{hw} k{ks} x{nprob} [{label}] {sel[0][1]} workgroups; us since the first wave's entry (median / max over workgroups, mean over {len(sel)} launches)")
        for i, nm in enumerate(NAMES):
            a, b = w0[:, :, i], w7[:, :, i]
            print(f"    {nm:15s} wave0 {np.median(a, 1).mean():6.2f} / {a.max(1).mean():6.2f}    wave7 {np.median(b, 1).mean():6.2f} / {b.max(1).mean():6.2f}")
