#!/usr/bin/env python3
"""Host time vs GPU time of a train step: how long does Python need to QUEUE a step (no synchronisation inside the loop),
and which part of train_step takes it?  (the GPU idles ~1.2 ms per step: tools/train_timeline.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import idealnerf_amd
from idealnerf_amd import synthetic, train as T_, ops
from idealnerf_amd.audio_exp_nerf import Network
from idealnerf_amd.helper import RenderConfig
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
N = 10
t0 = time.perf_counter()
marks = []
for i in range(N):
    T_.train_step(net, opt, data, latent_codes, 4 + i, 8)
    marks.append(time.perf_counter())
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host queues a step in {t_host / N * 1e3:.2f} ms (per step: {[round((b - a) * 1e3, 1) for a, b in zip([t0] + marks[:-1], marks)]}); GPU finishes in {t_all / N * 1e3:.2f} ms per step")
# where the host time goes: the same step under cProfile (host side only)
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    T_.train_step(net, opt, data, latent_codes, 20 + i, 8)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
