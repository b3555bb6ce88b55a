#!/usr/bin/env python3
"""Which host calls are behind the small launches of a train step (fills, copies, tiny elementwise kernels)?
One step under torch.profiler with Python stacks: every aten op that launches a kernel, grouped by (op, innermost
idealnerf_amd / torch.optim frame).  Prints counts per step.   python tools/train_small_ops.py"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import idealnerf_amd
from idealnerf_amd import synthetic, train as T_, ops
from idealnerf_amd.audio_exp_nerf import Network
from idealnerf_amd.helper import RenderConfig
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(0)
H = W = 450
syn = synthetic.frame(H, W, seed=0)
cfg = RenderConfig(perturb=1.0, chunk=8192, near=syn["near"], far=syn["far"])
net = Network(H, W, syn["focal"], syn["near"], syn["far"], 8192, None, 64, 128, args=cfg).to(dev).train()
synthetic.xavier_state_dict(net.face_nerf_coarse, 2, 300.0, 0.3)
synthetic.xavier_state_dict(net.face_nerf_fine, 3, 300.0, 0.3)
rs = np.random.RandomState(0)
sel = torch.from_numpy(rs.choice(H * W, 3072, replace=False))
rec = ops.frame_rays(syn["c2w"], H, W, syn["focal"], syn["near"], syn["far"], device=dev)
batch_rays = torch.stack([rec[sel.to(dev), 0:3], rec[sel.to(dev), 3:6]], 0).contiguous()
bg = syn["bc"].reshape(-1, 3)[sel].contiguous().to(dev)
tgt = torch.from_numpy(rs.uniform(0, 1, size=(len(sel), 3)).astype(np.float32)).to(dev)
auds = torch.from_numpy(rs.standard_normal((8, 16, 29)).astype(np.float32)).to(dev)
pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0).to(dev)
latent_codes = torch.ones(8, 32, device=dev, requires_grad=True)
opt = T_.make_optimizer(net, latent_codes)
data = (batch_rays[None], tgt, bg, auds[None], torch.zeros(1, H, W, 3), pose, syn["expr"][None].to(dev), torch.tensor([3]))
for i in range(4):
    T_.train_step(net, opt, data, latent_codes, i, 8)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(N):
        T_.train_step(net, opt, data, latent_codes, 4 + i, 8)
    torch.cuda.synchronize()
ev = prof.events()
launches = collections.Counter()
for e in ev:
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    kids = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not kids:
        continue
    # only leaf ops (the op that launched the kernel, not its callers)
    if any(c.kernels for c in e.cpu_children if hasattr(c, "kernels")):
        continue
    where = "?"
    for fr in (e.stack or []):
        if "ideal-nerf_amd" in fr or "torch/optim" in fr or "bench.py" in fr or "train_small_ops" in fr or "autograd/" in fr:
            where = fr.strip()[-110:]
            break
    launches[(e.name, where)] += len(kids)
tot = 0
for (name, where), n in sorted(launches.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:6.1f} per step  {name:34s} {where}")
    tot += n
print(f"{tot / N:.1f} kernel launches per step from aten ops")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
