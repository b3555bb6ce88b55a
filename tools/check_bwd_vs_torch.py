#!/usr/bin/env python3
"""Backward kernels vs an fp64 torch backprop through the SAME saved activations (masks taken
from the kernel's own activation slab), for the head coarse pass of the torso test instance."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import idealnerf_amd, oracle
from idealnerf_amd import ops, autograd as ag
from idealnerf_amd.helper import linspace01
import test_hip_parity as T

dev = torch.device("cuda:0")
net, syn, P, dims, d = T._torso_setup(idealnerf_amd, dev)
coarse = net.face_nerf_coarse
S = 64
with torch.no_grad():
    aud = net.aud_net(d["auds"][1:2].to(dev)).contiguous()
expr, lat = d["expr"].to(dev), d["latent"].to(dev)
rec = oracle.ray_records(d["batch_rays"][0], d["batch_rays"][1], T.NEAR, T.FAR).to(dev)
bc = d["bg"].to(dev).contiguous()
z = ops.coarse_depths(rec, linspace01(S, dev))
folded = coarse.folded_bias(aud, expr, lat)
raw, acts = ag._train_query(coarse, folded, rec, z)
n = rec.shape[0]; Pn = n * S; Pp = (Pn + 127) // 128 * 128
g_rgb = torch.randn(n, 3, device=dev) * 0.01
d_aud, d_lat = torch.zeros_like(aud), torch.zeros_like(lat)
grads = ag._pass_bwd(coarse, aud, expr, lat, acts, raw, z, rec, bc, g_rgb, None, None, None, d_aud, d_lat)
torch.cuda.synchronize()
# ---- fp64 reference on the same activations
A = acts.view(2560, -1)  # not a real view of matrices; slice by offsets instead
off = [0, 64, 128] + [128 + 256 * i for i in range(1, 9)]
def mat(o, w): return acts[o * Pp:(o + w) * Pp].view(Pp, w)[:Pn].double()
x0, dirs = mat(0, 64), mat(64, 64)
a = [mat(128 + 256 * i, 256) for i in range(8)]           # a1..a8
v = [mat(128 + 2048 + 128 * i, 128) for i in range(3)]    # v1..v3
raw64 = raw.double().requires_grad_(True)
comp = oracle.composite(raw64.cpu(), z.double().cpu(), rec[:, 3:6].double().cpu(), bc.double().cpu())
(comp[0] * g_rgb.double().cpu()).sum().backward()
d_raw = raw64.grad.to(dev).view(Pn, 4)
sd = {k: p.detach().double() for k, p in coarse.named_parameters()}
C = 64 + 79 + 32
cond = torch.cat([aud.double(), expr.double() / 3, lat.double()])
G = {}
d_rgb, d_sig = d_raw[:, :3], d_raw[:, 3:4]
G["rgb_linear.weight"] = d_rgb.t() @ v[2]; G["rgb_linear.bias"] = d_rgb.sum(0)
dl = (d_rgb @ sd["rgb_linear.weight"]) * (v[2] > 0)
for i in (2, 1):
    G[f"views_linears.{i}.weight"] = dl.t() @ v[i - 1]; G[f"views_linears.{i}.bias"] = dl.sum(0)
    dl = (dl @ sd[f"views_linears.{i}.weight"]) * (v[i - 1] > 0)
inp_v0 = torch.cat([a[7], dirs[:, :27], (expr.double() / 3)[None].expand(Pn, -1)], 1)
G["views_linears.0.weight"] = dl.t() @ inp_v0; G["views_linears.0.bias"] = dl.sum(0)
G["alpha_linear.weight"] = d_sig.t() @ a[7]; G["alpha_linear.bias"] = d_sig.sum(0)
dh = (dl @ sd["views_linears.0.weight"][:, :256] + d_sig @ sd["alpha_linear.weight"]) * (a[7] > 0)
for l in range(7, 0, -1):
    inp = a[l - 1] if l != 5 else torch.cat([x0[:, :63], cond[None].expand(Pn, -1), a[4]], 1)
    G[f"pts_linears.{l}.weight"] = dh.t() @ inp; G[f"pts_linears.{l}.bias"] = dh.sum(0)
    W = sd[f"pts_linears.{l}.weight"]
    W = W[:, 63 + C:] if l == 5 else W
    dh = (dh @ W) * (a[l - 1] > 0)
inp0 = torch.cat([x0[:, :63], cond[None].expand(Pn, -1)], 1)
G["pts_linears.0.weight"] = dh.t() @ inp0; G["pts_linears.0.bias"] = dh.sum(0)
errs = sorted(((float((grads[k].double() - G[k]).abs().max() / G[k].abs().max()), k) for k in G), reverse=True)
print("kernel backward vs fp64 torch on the same activations; worst:", [(f"{e:.1e}", k) for e, k in errs[:6]])
