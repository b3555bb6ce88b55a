"""Debug aid: plain-bf16 FaceNeRF on a few sizes, error per wave-sized group of rows vs the CPU oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import idealnerf_amd as idn
import oracle.render_oracle as o
dev = torch.device("cuda")
dims = o.facenerf_dims()
params = o.xavier_facenerf_params(11, dims)
rs = np.random.RandomState(0)
aud, expr, lat = (torch.from_numpy(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
sd = {k: t.to(dev).contiguous() for k, t in params.items()}
ps = idn.ops.params_struct(sd, 64, 76, 32)
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
packed = idn.ops.pack_weights(ps, dev, prec)
folded = idn.ops.fold_conditioning(ps, aud.to(dev), expr.to(dev), lat.to(dev), dev)
for n in (32, 128, 256, 512, 1024, 70000):
    x = torch.from_numpy(rs.uniform(-1, 1, size=(n, 90)).astype(np.float32))
    with torch.no_grad():
        ref = o.facenerf_forward(params, x, aud, expr, lat, dims).numpy()
    out = idn.ops.facenerf_fwd(packed, folded, x.to(dev), prec).cpu().numpy()
    scale = np.abs(ref).max()
    per = [np.abs(out[i:i + 32] - ref[i:i + 32]).max() / scale for i in range(0, min(n, 1024), 32)]
    print(n, "max rel err %.3e" % (np.abs(out - ref).max() / scale), "per 32 rows:", " ".join("%.0e" % e for e in per[:32]), flush=True)
