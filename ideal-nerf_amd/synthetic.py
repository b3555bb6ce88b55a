"""Seeded synthetic workload of the shape BASELINE.json names:
there is no dataset or checkpoint to load, so benchmarks and smoke tests render a
512x512 frame from a seeded pose with May near/far (configs/audio_expr_nerf/may/
paper_model/torso_bg.txt), uniform background, Gaussian audio/expression latents, the
trainer's ones latent code (audio_exp_nerf.py:482) and Xavier-uniform weights with bias
0.01 (audio_exp_nerf.py:442-448).  Pure numpy/torch data generation; no rendering here.
"""
import math

import numpy as np
import torch

NEAR, FAR = 0.5772005200386048, 1.1772005200386046


def frame(H=512, W=512, seed=0, dim_aud=64, dim_expr=76, dim_latent=32):
    rs = np.random.RandomState(seed)
    ang = rs.uniform(-0.08, 0.08, size=3)
    c, s = np.cos(ang), np.sin(ang)
    Rx = np.array([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
    Ry = np.array([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
    Rz = np.array([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
    c2w = np.concatenate([Rz @ Ry @ Rx, np.array([[0.0], [0.0], [0.877]])], axis=1).astype(np.float32)
    bc = np.random.RandomState(seed + 1).uniform(0, 1, size=(H, W, 3)).astype(np.float32)
    rs2 = np.random.RandomState(seed + 100)
    t = lambda a: None if a is None else torch.from_numpy(a)
    aud = rs2.standard_normal(dim_aud).astype(np.float32) if dim_aud else None
    expr = rs2.standard_normal(dim_expr).astype(np.float32) if dim_expr else None
    latent = np.ones(dim_latent, dtype=np.float32) if dim_latent else None
    return dict(H=H, W=W, focal=1200.0 * W / 450.0, c2w=torch.from_numpy(c2w), near=NEAR, far=FAR,
                bc=torch.from_numpy(bc), aud=t(aud), expr=t(expr), latent=t(latent))


def xavier_state_dict(module, seed, sigma_gain=None, sigma_bias=None):
    """Fill a FaceNeRF's parameters from numpy RandomState(seed) in state_dict order
    (Xavier-uniform weights, bias 0.01).  ``sigma_gain`` scales the density head so the
    volume is not empty (Xavier-initialised sigma is ~0 everywhere)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k, v in module.state_dict().items():
        if k.endswith(".weight"):
            bound = math.sqrt(6.0 / (v.shape[0] + v.shape[1]))
            sd[k] = torch.from_numpy(rs.uniform(-bound, bound, size=tuple(v.shape)).astype(np.float32))
        else:
            sd[k] = torch.full(tuple(v.shape), 0.01, dtype=torch.float32)
    if sigma_gain is not None:
        sd["alpha_linear.weight"] = sd["alpha_linear.weight"] * sigma_gain
    if sigma_bias is not None:
        sd["alpha_linear.bias"] = torch.full_like(sd["alpha_linear.bias"], sigma_bias)
    module.load_state_dict(sd)
    return module
