"""Render-math helpers with the reference's names and signatures
(NeRFs/HeadNeRF/helper.py:148-313, NeRFs/HeadNeRF/train/baseline.py:325-375), computed by
libidealnerf.so -- and the reference's flag surface: ``config_parser()`` (helper.py:16-138) and the module
attributes ``parser`` / ``args`` (helper.py:141-142), which upstream fills by parsing ``sys.argv`` when the module
is imported and which are filled here on first access (``from idealnerf_amd.helper import *`` IS a first access,
so the import swap keeps the import-time semantics).  A ``Network`` constructed without ``args=`` reads the
process's flags from there (config.default_render_config); ``RenderConfig`` is the explicit form of the same subset.
"""
from dataclasses import dataclass

import os
import sys

import einops
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


@dataclass
class RenderConfig:
    """The subset of the reference's flags the per-ray path reads (helper.py:16-138)."""
    netdepth: int = 8
    netwidth: int = 256
    dim_aud: int = 64
    dim_expr: int = 76
    dim_latent: int = 32
    win_size: int = 16
    smo_size: int = 8
    nosmo_iters: int = 300000
    N_samples: int = 64
    N_importance: int = 128
    perturb: float = 1.0          # reference default (helper.py:70); eval runs pass 0
    chunk: int = 1024 * 8
    netchunk: int = 1024 * 64
    multires: int = 10
    multires_views: int = 4
    use_viewdirs: bool = True
    near: float = 0.3
    far: float = 0.9
    lc_weight: float = 0.0005


def config_parser():
    """helper.py:16-138 -> a parser whose `parse_args()` resolves defaults <- `--config` file <- command line."""
    from .config import ConfigParser
    return ConfigParser("head")


def write_config(args):
    """helper.py:371-384."""
    from .config import write_config as _write
    return _write(args)


def __getattr__(name):
    """`helper.parser` / `helper.args`: upstream's import-time `parser = config_parser(); args = parser.parse_args()`
    (helper.py:141-142), evaluated on first access and then kept as ordinary module attributes."""
    if name in ("args", "parser"):
        g = globals()
        if "parser" not in g:
            g["parser"] = config_parser()
        if name == "args":
            g["args"] = g["parser"].parse_args()
        return g[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["RenderConfig", "config_parser", "write_config", "parser", "args", "img2mse", "mse2psnr", "to8b", "to8b_tensor",
           "linspace01", "get_embedder", "get_rays", "raw2outputs", "sample_pdf", "draw_sigma_noise",
           # upstream's scripts take these from `from NeRFs.HeadNeRF.helper import *` as well (helper.py:2-7):
           "np", "torch", "nn", "F", "einops", "os", "sys"]


def img2mse(x, y):
    return torch.nn.functional.mse_loss(x, y)


_log10_cache = {}


def mse2psnr(x):
    """helper.py:151: -10 log(x) / log(Tensor([10.])).  The constant is built once per device: creating it from a
    Python list on every call is a blocking host-to-device copy, i.e. a full stream synchronisation in the middle
    of every training step (the host could not queue the backward while the forward ran)."""
    c = _log10_cache.get(x.device)
    if c is None:
        c = _log10_cache[x.device] = torch.log(torch.tensor([10.0])).to(x.device)
    return -10.0 * torch.log(x) / c


def to8b(x):
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def to8b_tensor(x):
    """helper.py:157."""
    return 255 * torch.clip(x, 0, 1)


_linspace_cache = {}


def linspace01(n: int, device) -> torch.Tensor:
    """torch.linspace(0, 1, n) evaluated on the CPU exactly as the reference does
    (audio_exp_nerf.py:306, helper.py:280) and cached on the device: its roundings are not
    re-derivable in a kernel and they feed index decisions (DESIGN.md "numerics")."""
    key = (n, str(device))
    t = _linspace_cache.get(key)
    if t is None:
        t = torch.linspace(0.0, 1.0, steps=n).to(device)
        _linspace_cache[key] = t
    return t


def get_embedder(multires, i=0, input_dims=3):
    """helper.py:207-224.  The per-point encodings are fused into the MLP kernel; this
    standalone embedder only serves per-frame vectors (torso pose signal,
    train_torso.py:238-240) and is plain tensor math on whatever device x lives on."""
    if i == -1:
        return torch.nn.Identity(), input_dims
    out_dim = input_dims * (1 + 2 * multires)

    def embed(x):
        feats = [x]
        for b in range(multires):
            f = float(2 ** b)
            feats += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(feats, -1)

    return embed, out_dim


def get_rays(H, W, focal, c2w, cx=None, cy=None, near=0.0, far=1.0, row0=0, nrows=None, device="cuda"):
    """helper.py:228-243 on the GPU.  Returns (rays_o, rays_d) [nrows, W, 3]."""
    rec = ops.frame_rays(c2w.detach().cpu(), H, W, focal, near, far, row0, nrows, cx, cy, device)
    nrows = H - row0 if nrows is None else nrows
    return rec[:, 0:3].reshape(nrows, W, 3), rec[:, 3:6].reshape(nrows, W, 3)


def _as_records(rays_d, z_like):
    """raw2outputs only needs |d|: build ray records with d filled in."""
    n = rays_d.shape[0]
    rec = torch.zeros((n, 11), dtype=torch.float32, device=rays_d.device)
    rec[:, 3:6] = rays_d
    return rec


def draw_sigma_noise(shape, raw_noise_std, pytest, device):
    """The density noise of raw2outputs as the reference draws it (baseline.py:353-361): randn * std, or --
    under ``pytest`` -- numpy's UNIFORM rand(seed 0) * std (the reference's test override is uniform, not normal)."""
    if not raw_noise_std > 0.0:
        return None
    if pytest:
        np.random.seed(0)
        return torch.Tensor(np.random.rand(*list(shape)) * raw_noise_std).to(device)
    return torch.randn(tuple(shape), device=device) * raw_noise_std


def raw2outputs(raw, z_vals, rays_d, bc_rgb, raw_noise_std=0.0, white_bkgd=False, pytest=False):
    """baseline.py:325-375 -> (rgb_map, disp_map, acc_map, weights, depth_map)."""
    noise = draw_sigma_noise(raw.shape[:-1], raw_noise_std, pytest, raw.device)
    o = ops.composite_fwd(raw.contiguous(), z_vals.contiguous(), _as_records(rays_d, z_vals), bc_rgb.contiguous(),
                          sigma_noise=noise, white_bkgd=white_bkgd)
    return o["rgb_map"], o["disp_map"], o["acc_map"], o["weights"], o["depth_map"]


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """helper.py:269-313: bins [n, nb], weights [n, nb-1] -> samples [n, N_samples]."""
    dev = bins.device
    n, nb = bins.shape
    if u is None:
        if det and pytest:   # the reference's test override replaces torch.linspace by numpy's (helper.py:286-290):
            u = torch.Tensor(np.linspace(0., 1., N_samples)).to(dev)   # 8 of 128 values differ in the last bit
        elif det:
            u = linspace01(N_samples, dev)
        elif pytest:
            np.random.seed(0)
            u = torch.Tensor(np.random.rand(n, N_samples)).to(dev)
        else:
            u = torch.rand((n, N_samples), device=dev)
    # pdf, cdf and the inversion all run in the kernel Network.render_rays uses (torch.sum in ATen's CPU
    # order, fp64 cumsum): the standalone helper and the renderer agree bit for bit on cdf and indices
    o = ops.sample_pdf_bins_fwd(bins.to(torch.float32).contiguous(), weights.to(torch.float32).contiguous(),
                                u.to(torch.float32).contiguous())
    return o["z_samples"]
