import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, idealnerf_amd
from idealnerf_amd import synthetic, ops
from idealnerf_amd.helper import linspace01
dev = torch.device("cuda:0")
net = synthetic.xavier_state_dict(idealnerf_amd.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76), 3).to(dev)
syn = synthetic.frame(32, 32)
cond = [syn[k].to(dev) for k in ("aud", "expr", "latent")]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
net.precision = prec
pk = net.packed_weights(); fb = net.folded_bias(*cond)
rays = ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], syn["near"], syn["far"], 0, 2, device=dev)
z = ops.coarse_depths(rays, linspace01(64, dev)); torch.cuda.synchronize()
print("mlp rays", prec, flush=True)
raw = ops.query_rays_fwd(pk, fb, rays, z, net.prec_code); torch.cuda.synchronize()
print("done", float(raw.abs().mean()), bool(torch.isfinite(raw).all()))
