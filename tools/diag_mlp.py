#!/usr/bin/env python3
"""Where does a tile pass of the MLP kernel spend its cycles?  Runs the diagnostic build
(python ideal-nerf_amd/build.py --diag; IDN_LIB=.../libidealnerf_diag.so) on the bench
workload's fine pass and prints per-category shares of wave cycles.  Shares only -- the
stamps themselves cost cycles, so the diagnostic build's run time is not quoted."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("IDN_LIB", os.path.join(ROOT, "ideal-nerf_amd", "libidealnerf_diag.so"))
import torch
import idealnerf_amd
from idealnerf_amd import ops, synthetic
from idealnerf_amd.helper import linspace01

dev = torch.device("cuda:0")
lib = idealnerf_amd._lib.load()
lib.idealnerf_diag_read.argtypes = [C.POINTER(C.c_ulonglong)]
syn = synthetic.frame(512, 512, seed=0)
net = synthetic.xavier_state_dict(idealnerf_amd.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76), 3, 300.0, 0.3).to(dev)
g = lambda t: t.to(dev)
folded = net.folded_bias(g(syn["aud"]), g(syn["expr"]), g(syn["latent"]))
rays = ops.frame_rays(syn["c2w"], 512, 512, syn["focal"], syn["near"], syn["far"], 0, 64, device=dev)
z = ops.coarse_depths(rays, linspace01(192, dev))
buf = (C.c_ulonglong * 8)()
for rep in range(2):
    ops.query_rays_fwd(net.packed_weights(), folded, rays, z)
    torch.cuda.synchronize()
    lib.idealnerf_diag_read(buf)
tot = buf[0]
names = ["total", "input+PE", "barrier", "layer boundary (relu/bias)", "store"]
print("waves:", buf[5], " cycles per wave:", tot / max(buf[5], 1))
for i, n in enumerate(names):
    print(f"{n:28s} {buf[i] / tot * 100:6.2f} %")
print(f"{'MFMA stream (remainder)':28s} {(tot - sum(buf[1:5])) / tot * 100:6.2f} %")
