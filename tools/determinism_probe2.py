#!/usr/bin/env python3
"""Full head+torso training flow (tests/test_hip_parity.py::_torso_setup) run repeatedly in one
process: every gradient must be bitwise identical between runs."""
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
ref = None
for it in range(10):
    net.zero_grad(set_to_none=True)
    rgb_com, rgb_com0 = net([x, 0, 4])
    loss = ((rgb_com - tgt) ** 2).mean() + ((rgb_com0 - tgt) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    cur = {"rgb_com": rgb_com.detach().clone(), "rgb_com0": rgb_com0.detach().clone()}
    cur.update({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    if ref is None:
        ref = cur
    else:
        bad = {k: (int((cur[k] != ref[k]).sum()), float((cur[k] - ref[k]).abs().max() / ref[k].abs().max())) for k in cur if not torch.equal(cur[k], ref[k])}
        print(f"run {it}: differing:", bad if bad else "none")
