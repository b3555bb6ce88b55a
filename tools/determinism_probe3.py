#!/usr/bin/env python3
"""One fresh process = one head+torso train step; print the worst gradient errors vs the CPU
oracle and a checksum of the audio feature (cross-process variability hunt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import idealnerf_amd
import test_hip_parity as T

dev = torch.device("cuda:0")
net, syn, P, dims, d = T._torso_setup(idealnerf_amd, dev)
net.train()
x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
     d["expr"][None], d["latent"], torch.tensor([1]))
tgt = d["target"].to(dev)
with torch.no_grad():
    af = net.aud_net(d["auds"][1:2].to(dev))
rgb_com, rgb_com0 = net([x, 0, 4])
loss = ((rgb_com - tgt) ** 2).mean() + ((rgb_com0 - tgt) ** 2).mean()
loss.backward()
for p in P.values():
    for v in p.values():
        v.requires_grad_(True)
(ref, ref0), aud_net = T._torso_oracle(net, P, dims, d, grad=True)
loss_o = ((ref - d["target"]) ** 2).mean() + ((ref0 - d["target"]) ** 2).mean()
loss_o.backward()
errs = []
for tag, m in (("hc", net.face_nerf_coarse), ("tc", net.torso_coarse_nerf), ("hf", net.face_nerf_fine), ("tf", net.torso_fine_nerf)):
    for name, prm in m.named_parameters():
        if name.startswith("feature_linear"):
            continue
        errs.append((float(T.rel_err(prm.grad, P[tag][name].grad)), tag + "." + name))
errs.sort(reverse=True)
print("aud checksum %.9e  loss %.9e | worst:" % (float(af.double().sum()), float(loss)), [(f"{e:.1e}", n) for e, n in errs[:4]])
