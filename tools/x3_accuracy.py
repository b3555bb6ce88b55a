"""Diagnostic: network-output error of the x3 modes on the reference's golden FaceNeRF inputs, and their deviation
from the fp32 mode on the sharp head+torso scene (the importance-sampling amplification)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import idealnerf_amd as idn
import test_hip_parity as tp
import oracle
dev = torch.device("cuda")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "facenerf.npz")))
dims = oracle.facenerf_dims()
sd = {k: t.to(dev).contiguous() for k, t in oracle.xavier_facenerf_params(11, dims).items()}
ps = idn.ops.params_struct(sd, 64, 76, 32)
folded = idn.ops.fold_conditioning(ps, *(torch.from_numpy(g["c235_" + k]).to(dev) for k in ("aud", "expr", "latent")), dev)
for name, code in (("f32", 0), ("bf16x3", 1), ("fp16x3", 3)):
    out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, code), folded, torch.from_numpy(g["c235_x"]).to(dev), code)
    print(f"FaceNeRF golden  {name:7s} max rel err {tp.rel_err(out, g['c235_out']):.2e}")
net, syn, P, dims2, d = tp._torso_setup(idn, dev, n=512)
x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
     d["expr"][None], d["latent"], torch.tensor([1]))
net.train()
outs = {}
with torch.no_grad():
    for mode in ("f32", "fp16x3", "bf16x3", "mixed"):
        idn.set_render_precision(net, mode)
        outs[mode] = net([x, 0, 4])[0].double().cpu().numpy()
for mode in ("fp16x3", "bf16x3", "mixed"):
    e = np.abs(outs[mode] - outs["f32"]).max(1)
    print(f"head+torso scene {mode:7s} vs fp32: max {e.max():.2e}  rays > 1e-4: {(e > 1e-4).sum()} / {len(e)}  PSNR {-10*np.log10(max((e**2).mean(),1e-30)):.1f} dB")
