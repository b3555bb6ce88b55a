#!/usr/bin/env python3
"""Is the bf16x6 forward deterministic, and where does it differ from the fp32 kernel?  (debugging aid)
    IDN_LIB=... python tools/x6_determinism.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import idealnerf_amd as idn, oracle
dev = torch.device("cuda:0")
NEAR, FAR = 0.5772005200386048, 1.1772005200386046
for dims_kw, seed, nrays, S in ((dict(dim_aud=64, dim_expr=79, dim_latent=32), 22, 512, 192), (dict(dim_aud=106, dim_expr=0, dim_latent=0), 24, 512, 192),
                                (dict(dim_aud=64, dim_expr=76, dim_latent=32), 3, 4096, 64)):
    dims = oracle.facenerf_dims(**dims_kw)
    p = oracle.xavier_facenerf_params(seed, dims)
    p["alpha_linear.weight"] = p["alpha_linear.weight"] * 100.0
    sd = {k: v.to(dev).contiguous() for k, v in p.items()}
    ps = idn.ops.params_struct(sd, dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])
    rs = np.random.RandomState(1)
    cond = [None if not d else torch.from_numpy(rs.standard_normal(d).astype(np.float32)).to(dev) for d in (dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])]
    folded = idn.ops.fold_conditioning(ps, *cond, dev)
    syn = oracle.synthetic_frame(64, 64, seed=4)
    rays = idn.ops.frame_rays(syn["c2w"], 64, 64, syn["focal"], NEAR, FAR, device=dev)[:nrays].contiguous()
    z = idn.ops.coarse_depths(rays, torch.linspace(0, 1, S).to(dev))
    pk6, pk32 = idn.ops.pack_weights(ps, dev, 4), idn.ops.pack_weights(ps, dev, 0)
    ref = idn.ops.query_rays_fwd(pk32, folded, rays, z, 0)
    outs = [idn.ops.query_rays_fwd(pk6, folded, rays, z, 4).clone() for _ in range(6)]
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    for i, o in enumerate(outs):
        d = (o - ref).abs().reshape(-1, 4).max(1)[0] / scale
        same = bool(torch.equal(o, outs[0]))
        bad = torch.nonzero(d > 1e-4).flatten()
        print(f"dims {dims_kw} run {i}: max err vs f32 {float(d.max()):.2e}, identical to run 0: {same}, points > 1e-4: {bad.numel()}"
              + (f" first {bad[:8].tolist()} (tile {int(bad[0]) // 128}, wave {int(bad[0]) % 128 // 32}, lane {int(bad[0]) % 32})" if bad.numel() else ""))
