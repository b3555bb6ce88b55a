"""Diagnostic: where does bf16x3 differ from fp32 on the head+torso test scene? (max-norm vs rms, per output)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import idealnerf_amd as idn
import test_hip_parity as tp
dev = torch.device("cuda")
net, syn, P, dims, d = tp._torso_setup(idn, dev, n=512)
net.train()
outs = {}
kw = dict(H=32, W=32, focal=net.focal, render_poses=None, chunk=512, near=net.near, far=net.far, bc_rgb=d["bg"].to(dev))
aud = net.aud_net(d["auds"][1].unsqueeze(0).to(dev))
aud_t = net.torso_signal(aud, d["pose"].to(dev))
with torch.no_grad():
    for prec in ("f32", "bf16x3"):
        for m in (net.face_nerf_coarse, net.face_nerf_fine, net.torso_coarse_nerf, net.torso_fine_nerf):
            m.precision = prec
        h = net.render_pair(expr=d["expr"].to(dev), latent_code=d["latent"].to(dev), rays=d["batch_rays"].to(dev), aud_para=aud,
                            network_nerf={"coarse": net.face_nerf_coarse, "fine": net.face_nerf_fine}, **kw)
        t = net.render_pair(expr=None, latent_code=None, rays=d["batch_rays_torso"].to(dev), aud_para=aud_t,
                            network_nerf={"coarse": net.torso_coarse_nerf, "fine": net.torso_fine_nerf}, **kw)
        outs[prec] = dict(head_rgb=h[0], head_lw=h[3], torso_lw=t[3], torso_fg=t[4], com=h[0] * t[3][..., None] + t[4],
                          head_rgb0=h[5]["rgb0"], torso_zstd=t[5]["z_std"])
for k in outs["f32"]:
    a, b = outs["f32"][k].double().cpu().numpy(), outs["bf16x3"][k].double().cpu().numpy()
    e = np.abs(a - b).reshape(len(a), -1).max(1)
    print(f"{k:10s} max|ref| {np.abs(a).max():.3f}  max err {e.max():.2e}  rms {np.sqrt((e**2).mean()):.2e}  rays > 1e-4: {(e > 1e-4 * np.abs(a).max()).sum()} / {len(e)}  p99 {np.percentile(e, 99):.2e}")
