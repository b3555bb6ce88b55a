import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, idealnerf_amd
from idealnerf_amd import synthetic
dev = torch.device("cuda:0")
net = synthetic.xavier_state_dict(idealnerf_amd.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76), 3).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.rand(n, 90, device=dev)
cond = [torch.randn(k, device=dev) for k in (64, 76, 32)]
net.precision = "bf16x3"
print("packing", flush=True); pk = net.packed_weights(); torch.cuda.synchronize()
print("folding", flush=True); fb = net.folded_bias(*cond); torch.cuda.synchronize()
print("mlp", flush=True); out = idealnerf_amd.ops.facenerf_fwd(pk, fb, x, 1); torch.cuda.synchronize()
print("done", out[:2].cpu())
