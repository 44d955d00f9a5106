"""Which Python lines of the training step emit the launches that are not convolutions: every GPU kernel of one eager step
(torch profiler, with stacks), grouped by (kernel name, innermost mcquic_amd frame of the op that launched it).
    python tools/prof_train_glue.py            (on the GPU box)
"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import torch
from torch.profiler import profile, ProfilerActivity
from mcquic_amd import Compressor

dev = torch.device("cuda:0")
torch.manual_seed(3407)
model = Compressor(128, 2, [8192, 2048, 512]).to(dev).train()
x = (torch.rand((8, 3, 256, 256)) * 2 - 1).to(dev)


def step():
    for p in model.parameters():
        p.grad = None
    xHat, yHat, codes, logits = model(x)
    loss = torch.nn.functional.mse_loss(xHat, x)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

# kernels are linked to the CPU op that launched them through the correlation id of the runtime call
events = prof.profiler.kineto_results.events()
launch_of = {}          # correlation id -> (kernel name, duration us)
cpu_ops = []            # (start, end, name, stack)
for e in events:
    dt = str(e.device_type())
    if "CUDA" in dt or "PrivateUse" in dt or "HIP" in dt:
        launch_of.setdefault(e.correlation_id(), []).append((e.name(), e.duration_ns() / 1e3))
by_corr = {}
for e in events:
    if "CPU" in str(e.device_type()) and e.correlation_id() in launch_of and e.name().startswith(("hipLaunch", "hipExtLaunch", "hipMemcpy", "hipMemset", "hipModuleLaunch", "hipExtModuleLaunch")):
        by_corr[e.correlation_id()] = (e.start_ns(), e.name())
ops = [(e.start_ns(), e.start_ns() + e.duration_ns(), e.name(), e.stack()) for e in events if "CPU" in str(e.device_type()) and e.stack()]
ops.sort()


def frame_for(t):
    best = None
    for s, en, name, stack in ops:
        if s > t:
            break
        if en >= t:
            best = (name, stack)
    if best is None:
        return "?", "?"
    name, stack = best
    where = "?"
    for fr in stack:
        if "mcquic_amd" in fr or "tools/" in fr:
            where = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            where = where.split("mcquic_amd/")[-1] if "mcquic_amd/" in where else where
            break
    return name, where


agg = collections.defaultdict(lambda: [0, 0.0])
for corr, (t, api) in by_corr.items():
    opname, where = frame_for(t)
    for kname, dur in launch_of[corr]:
        k = kname.split("(")[0][-70:]
        if "conv_mfma" in k or "conv_wgrad" in k or "conv_t16" in k:
            continue
        a = agg[(k, opname[:40], where[:80])]
        a[0] += 1
        a[1] += dur
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
n = sum(v[0] for v in agg.values())
print(f"# non-conv launches of one training step: {n} launches, {tot / 1e3:.2f} ms of kernel time")
print(f"{'us':>9} {'n':>4}  kernel | op | frame")
for (k, o, w), (c, d) in rows[:120]:
    print(f"{d:9.1f} {c:4d}  {k} | {o} | {w}")

print("\n# aten / autograd ops by count (with their innermost frames)")
try:
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=50, max_src_column_width=90))
except Exception as e:   # noqa
    print("table failed:", e)
