#!/usr/bin/env python3
"""Run the same training forward/backward several times and report which tensors differ
bitwise between runs (the kernels use no float atomics: everything must be identical)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import idealnerf_amd
from idealnerf_amd import ops, synthetic, autograd as ag
from idealnerf_amd.helper import linspace01

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = synthetic.xavier_state_dict(idealnerf_amd.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=79), 21, 100.0, 0.2).to(dev)
syn = synthetic.frame(32, 32, seed=4, dim_expr=79)
g = lambda t: t.to(dev)
aud, expr, lat = g(syn["aud"]), g(syn["expr"]), g(syn["latent"])
rays_all = ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], syn["near"], syn["far"], device=dev)
sel = torch.from_numpy(np.random.RandomState(5).choice(1024, 48, replace=False)).to(dev)
rays = rays_all[sel].contiguous()
bc = g(syn["bc"]).reshape(-1, 3)[sel].contiguous()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
z = ops.coarse_depths(rays, linspace01(S, dev))
folded = net.folded_bias(aud, expr, lat)
g_rgb = torch.randn(48, 3, device=dev)
ref = None
for it in range(8):
    raw, acts = ag._train_query(net, folded, rays, z)
    # churn the allocator / caches between runs
    junk = torch.randn(1 << 22, device=dev)
    d_aud, d_lat = torch.zeros_like(aud), torch.zeros_like(lat)
    grads = ag._pass_bwd(net, aud, expr, lat, acts, raw, z, rays, bc, g_rgb, None, None, None, d_aud, d_lat)
    torch.cuda.synchronize()
    cur = {"raw": raw.clone(), "acts": acts.clone(), "d_aud": d_aud.clone(), "d_lat": d_lat.clone()}
    cur.update({"g." + k: v.clone() for k, v in grads.items()})
    if ref is None:
        ref = cur
        print("nan in grads:", {k: bool(torch.isnan(v).any()) for k, v in cur.items() if torch.isnan(v).any()})
    else:
        bad = {k: int((cur[k] != ref[k]).sum()) for k in cur if not torch.equal(cur[k], ref[k])}
        print(f"run {it}: differing tensors:", bad if bad else "none")
