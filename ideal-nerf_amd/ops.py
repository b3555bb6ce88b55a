"""Tensor-level entry points over the C ABI.  torch is plumbing here: device memory,
the current HIP stream and nothing else.  Every function requires CUDA (ROCm) fp32
contiguous tensors and raises otherwise -- no silent fallback.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import IDN_PREC_F32, IdealNerfError, check

PTS_CH, VIEWS_CH, W_HID = 63, 27, 256


def _ptr(t: Optional[torch.Tensor], name="tensor", dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise IdealNerfError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise IdealNerfError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise IdealNerfError(f"{name} must be contiguous")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def params_struct(sd: Dict[str, torch.Tensor], dim_aud: int, dim_expr: int, dim_latent: int, prefix: str = ""):
    """state_dict-keyed tensors (models/face_nerf.py:27-37 names) -> idn_facenerf_params."""
    p = _lib.FaceNerfParams()
    C_all = PTS_CH + dim_aud + dim_expr + dim_latent

    def get(key, shape):
        t = sd[prefix + key]
        if tuple(t.shape) != shape:
            raise IdealNerfError(f"{key}: shape {tuple(t.shape)} != {shape} "
                                 "(only D=8, W=256, skips=[4], multires=10/4 is compiled)")
        return _ptr(t, key)

    for i in range(8):
        fan_in = C_all if i == 0 else (W_HID + C_all if i == 5 else W_HID)
        p.pts_w[i] = get(f"pts_linears.{i}.weight", (W_HID, fan_in))
        p.pts_b[i] = get(f"pts_linears.{i}.bias", (W_HID,))
    p.views_w[0] = get("views_linears.0.weight", (W_HID // 2, W_HID + VIEWS_CH + dim_expr))
    p.views_b[0] = get("views_linears.0.bias", (W_HID // 2,))
    for i in (1, 2):
        p.views_w[i] = get(f"views_linears.{i}.weight", (W_HID // 2, W_HID // 2))
        p.views_b[i] = get(f"views_linears.{i}.bias", (W_HID // 2,))
    p.alpha_w = get("alpha_linear.weight", (1, W_HID))
    p.alpha_b = get("alpha_linear.bias", (1,))
    p.rgb_w = get("rgb_linear.weight", (3, W_HID // 2))
    p.rgb_b = get("rgb_linear.bias", (3,))
    p.dim_aud, p.dim_expr, p.dim_latent = dim_aud, dim_expr, dim_latent
    return p


def pack_weights(p, device, precision=IDN_PREC_F32) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty(lib.idealnerf_packed_weight_floats(precision), dtype=torch.float32, device=device)
    check(lib.idealnerf_pack_weights(C.byref(p), precision, out.data_ptr(), _stream()))
    return out


def fold_conditioning(p, aud, expr, latent, device) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty(lib.idealnerf_folded_bias_floats(), dtype=torch.float32, device=device)
    check(lib.idealnerf_fold_conditioning(C.byref(p), _ptr(aud, "aud"), _ptr(expr, "expr"), _ptr(latent, "latent"),
                                          out.data_ptr(), _stream()))
    return out


def facenerf_fwd(packed, folded, x, precision=IDN_PREC_F32) -> torch.Tensor:
    lib = _lib.load()
    if x.dim() != 2 or x.shape[1] != PTS_CH + VIEWS_CH:
        raise IdealNerfError(f"x must be [N, {PTS_CH + VIEWS_CH}], got {tuple(x.shape)}")
    out = torch.empty((x.shape[0], 4), dtype=torch.float32, device=x.device)
    check(lib.idealnerf_facenerf_fwd(_ptr(packed), _ptr(folded), precision, _ptr(x, "x"), x.shape[0],
                                     out.data_ptr(), _stream()))
    return out


def query_rays_fwd(packed, folded, rays, z, precision=IDN_PREC_F32) -> torch.Tensor:
    lib = _lib.load()
    n, S = z.shape
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=z.device)
    check(lib.idealnerf_query_rays_fwd(_ptr(packed), _ptr(folded), precision, _ptr(rays, "rays"), _ptr(z, "z"), n, S,
                                       raw.data_ptr(), _stream()))
    return raw


def query_points_fwd(packed, folded, pts, viewdirs, precision=IDN_PREC_F32) -> torch.Tensor:
    lib = _lib.load()
    n, S, _ = pts.shape
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=pts.device)
    check(lib.idealnerf_query_points_fwd(_ptr(packed), _ptr(folded), precision, _ptr(pts, "pts"),
                                         _ptr(viewdirs, "viewdirs"), n, S, raw.data_ptr(), _stream()))
    return raw


def frame_rays(c2w, H, W, focal, near, far, row0=0, nrows=None, cx=None, cy=None, device="cuda") -> torch.Tensor:
    lib = _lib.load()
    if torch.device(device).type != "cuda":
        raise IdealNerfError(f"frame_rays renders on the GPU (got device {device}); the HIP path has no CPU fallback")
    nrows = H - row0 if nrows is None else nrows
    m = (C.c_float * 12)(*[float(v) for v in c2w[:3, :4].reshape(-1).tolist()])
    out = torch.empty((nrows * W, _lib.RAY_FLOATS), dtype=torch.float32, device=device)
    check(lib.idealnerf_frame_rays(m, H, W, float(focal), -1.0 if cx is None else float(cx),
                                   -1.0 if cy is None else float(cy), float(near), float(far), row0, nrows,
                                   out.data_ptr(), _stream()))
    return out


def to8b(rgb, swap_rb=False, nonfinite_flag=None) -> torch.Tensor:
    """helper.py:154 on the device: [..., 3] fp32 -> [..., 3] uint8.  `nonfinite_flag` (int32[1] on
    the device) is OR-ed with 1 when the frame holds a NaN/Inf."""
    lib = _lib.load()
    if rgb.shape[-1] != 3:
        raise IdealNerfError(f"to8b expects [..., 3], got {tuple(rgb.shape)}")
    out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
    if nonfinite_flag is not None and (nonfinite_flag.dtype != torch.int32 or not nonfinite_flag.is_cuda):
        raise IdealNerfError("nonfinite_flag must be an int32 device tensor")
    check(lib.idealnerf_to8b(_ptr(rgb, "rgb"), rgb.numel() // 3, int(bool(swap_rb)), out.data_ptr(),
                             nonfinite_flag.data_ptr() if nonfinite_flag is not None else None, _stream()))
    return out


def coarse_depths(rays, t_vals, t_rand=None) -> torch.Tensor:
    lib = _lib.load()
    n, S = rays.shape[0], t_vals.shape[0]
    z = torch.empty((n, S), dtype=torch.float32, device=rays.device)
    check(lib.idealnerf_coarse_depths(_ptr(rays, "rays"), _ptr(t_vals, "t_vals"), _ptr(t_rand, "t_rand"), n, S,
                                      z.data_ptr(), _stream()))
    return z


def composite_fwd(raw, z, rays, bc_rgb, with_fg=False, with_weights=True) -> Dict[str, torch.Tensor]:
    lib = _lib.load()
    n, S = z.shape
    dev = z.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    o = dict(rgb_map=new(n, 3), disp_map=new(n), acc_map=new(n), depth_map=new(n), last_weight=new(n))
    if with_weights:
        o["weights"] = new(n, S)
    if with_fg:
        o["rgb_fg"] = new(n, 3)
    co = _lib.CompositeOut(**{k: v.data_ptr() for k, v in o.items()})
    check(lib.idealnerf_composite_fwd(_ptr(raw, "raw"), _ptr(z, "z"), _ptr(rays, "rays"), _ptr(bc_rgb, "bc_rgb"), n, S,
                                      C.byref(co), _stream()))
    return o


def sample_pdf_fwd(z, weights, u, n_importance) -> Dict[str, torch.Tensor]:
    lib = _lib.load()
    n, S = z.shape
    dev = z.device
    per_ray = 1 if u.dim() == 2 else 0
    o = dict(z_samples=torch.empty((n, n_importance), dtype=torch.float32, device=dev),
             inds=torch.empty((n, n_importance), dtype=torch.int64, device=dev),
             cdf=torch.empty((n, S - 1), dtype=torch.float32, device=dev),
             z_fine=torch.empty((n, S + n_importance), dtype=torch.float32, device=dev),
             z_std=torch.empty((n,), dtype=torch.float32, device=dev))
    check(lib.idealnerf_sample_pdf_fwd(_ptr(z, "z"), _ptr(weights, "weights"), _ptr(u, "u"), per_ray, n, S,
                                       n_importance, o["z_samples"].data_ptr(), o["inds"].data_ptr(),
                                       o["cdf"].data_ptr(), o["z_fine"].data_ptr(), o["z_std"].data_ptr(), _stream()))
    return o


def invert_cdf(cdf, bins, u):
    lib = _lib.load()
    n, nb = cdf.shape
    per_ray = 1 if u.dim() == 2 else 0
    ni = u.shape[-1]
    zs = torch.empty((n, ni), dtype=torch.float32, device=cdf.device)
    inds = torch.empty((n, ni), dtype=torch.int64, device=cdf.device)
    check(lib.idealnerf_invert_cdf(_ptr(cdf, "cdf"), _ptr(bins, "bins"), _ptr(u, "u"), per_ray, n, nb, ni,
                                   zs.data_ptr(), inds.data_ptr(), _stream()))
    return zs, inds


_workspaces: Dict[tuple, torch.Tensor] = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def render_rays_fwd(rays, bc_rgb, packed_c, folded_c, packed_f, folded_f, t_vals, u, n_importance,
                    t_rand=None, with_fg=False, taps=False, precision=IDN_PREC_F32, precision_fine=None) -> Dict[str, torch.Tensor]:
    """Network.render_rays forward (audio_exp_nerf.py:297-371) as one C call.  `precision_fine` (default: the
    same as `precision`) selects the fine network's arithmetic; packed_f must be packed for it."""
    lib = _lib.load()
    n, S, Ni = rays.shape[0], t_vals.shape[0], int(n_importance)
    dev = rays.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    out = dict(rgb_map=new(n, 3), disp_map=new(n), acc_map=new(n))
    if Ni > 0:
        out.update(rgb0=new(n, 3), disp0=new(n), acc0=new(n), z_std=new(n), last_weight=new(n))
    if with_fg:
        out["rgb_fg"] = new(n, 3)
        if Ni > 0:
            out.update(rgb_fg0=new(n, 3), last_weight0=new(n))
    if taps:
        out.update(tap_z_coarse=new(n, S), tap_raw_coarse=new(n, S, 4), tap_weights_coarse=new(n, S))
        if Ni > 0:
            out.update(tap_cdf=new(n, S - 1), tap_inds=torch.empty((n, Ni), dtype=torch.int64, device=dev),
                       tap_z_samples=new(n, Ni), tap_z_fine=new(n, S + Ni), tap_raw_fine=new(n, S + Ni, 4),
                       tap_weights_fine=new(n, S + Ni))
    nbytes = lib.idealnerf_render_workspace_bytes(n, S, Ni)
    ws = _workspace(nbytes, dev)
    a = _lib.RenderArgs()
    a.rays, a.bc_rgb, a.n_rays = _ptr(rays, "rays"), _ptr(bc_rgb, "bc_rgb"), n
    a.n_samples, a.n_importance, a.precision = S, Ni, precision
    a.precision_fine_plus1 = 0 if precision_fine is None else int(precision_fine) + 1
    a.packed_coarse, a.folded_coarse = _ptr(packed_c), _ptr(folded_c)
    a.packed_fine, a.folded_fine = _ptr(packed_f), _ptr(folded_f)
    a.t_vals, a.t_rand = _ptr(t_vals, "t_vals"), _ptr(t_rand, "t_rand")
    a.u = _ptr(u, "u") if u is not None else None
    a.u_per_ray = 1 if (u is not None and u.dim() == 2) else 0
    for k, v in out.items():
        setattr(a, k, v.data_ptr())
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib.idealnerf_render_rays_fwd(C.byref(a), _stream()))
    return out
