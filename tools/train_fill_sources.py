#!/usr/bin/env python3
"""Who asks for the fill / copy launches of a train step?  A TorchDispatchMode logs every aten fill_ / zero_ / zeros / full /
copy_ / clone that touches a GPU tensor during two steps, with the innermost Python frame of this repository (or of torch.optim /
the autograd engine) on the stack.   python tools/train_fill_sources.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
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
WATCH = ("fill_", "zero_", "zeros", "zeros_like", "full", "full_like", "ones", "ones_like", "copy_", "clone", "_to_copy", "new_zeros", "empty_like")
log = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = out if torch.is_tensor(out) else (args[0] if args and torch.is_tensor(args[0]) else None)
            if t is not None and t.is_cuda:
                where = "(no Python frame: autograd engine / C++)"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    f = fr.filename
                    if "ideal-nerf_amd" in f or "torch/optim" in f or f.endswith("train_fill_sources.py"):
                        where = f"{os.path.basename(f)}:{fr.lineno} {fr.name}"
                        break
                log[(name, tuple(t.shape), where)] += 1
        return out


N = 2
with Spy():
    for i in range(N):
        T_.train_step(net, opt, data, latent_codes, 4 + i, 8)
torch.cuda.synchronize()
tot = 0
for (name, shape, where), n in sorted(log.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:5.1f} per step  {name:12s} {str(list(shape)):18s} {where}")
    tot += n
print(f"{tot / N:.1f} watched aten calls on GPU tensors per step")
