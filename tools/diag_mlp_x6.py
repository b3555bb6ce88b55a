#!/usr/bin/env python3
"""Where does a pass of the bf16x6 MLP kernel spend its cycles -- inference and the training (activation-saving)
variant?  Runs the diagnostic build (python ideal-nerf_amd/build.py --diag; IDN_LIB=.../libidealnerf_diag.so) on the
train step's fine pass (3072 rays x 192 points) and prints per-category shares of wave cycles: input + encoding,
slice barriers (the counted vmcnt wait + s_barrier), layer ends (record + convert), the rest = the MFMA phases.
Shares only -- the stamps themselves cost cycles."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("IDN_LIB", os.path.join(ROOT, "ideal-nerf_amd", "libidealnerf_diag.so"))
import torch
import idealnerf_amd
from idealnerf_amd import autograd, ops, synthetic
from idealnerf_amd.helper import linspace01

dev = torch.device("cuda:0")
lib = idealnerf_amd._lib.load()
lib.idealnerf_diag_read_x6.argtypes = [C.POINTER(C.c_ulonglong)]
syn = synthetic.frame(512, 512, seed=0)
net = synthetic.xavier_state_dict(idealnerf_amd.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76), 3, 300.0, 0.3).to(dev)
g = lambda t: t.to(dev)
folded = net.folded_bias(g(syn["aud"]), g(syn["expr"]), g(syn["latent"]))
rays = ops.frame_rays(syn["c2w"], 512, 512, syn["focal"], syn["near"], syn["far"], 0, 6, device=dev)   # 3072 rays
z = ops.coarse_depths(rays, linspace01(192, dev))
buf = (C.c_ulonglong * 8)()
names = ["total", "input + encoding", "slice barriers", "layer ends (record, convert)", "store"]
for what in ("inference", "training (saves activations)"):
    for rep in range(2):
        if what == "inference":
            ops.query_rays_fwd(net.packed_weights("bf16x6"), folded, rays, z, idealnerf_amd._lib.IDN_PREC_BF16X6)
        else:
            autograd._train_query(net, folded, rays, z)
        torch.cuda.synchronize()
        lib.idealnerf_diag_read_x6(buf)
    tot = buf[0]
    print(f"bf16x6 {what}: waves {buf[5]}, cycles per wave {tot / max(buf[5], 1):.0f}")
    for i in (1, 2, 3):
        print(f"  {names[i]:30s} {buf[i] / tot * 100:6.2f} %")
    print(f"  {'MFMA phases (remainder)':30s} {(tot - sum(buf[1:4])) / tot * 100:6.2f} %")
