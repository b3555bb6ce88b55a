"""Tensor-level entry points over the C ABI.  torch is plumbing here: device memory,
the HIP stream of the tensors' device and nothing else.  Every function requires CUDA
(ROCm) fp32 contiguous tensors of the documented shapes, all on ONE device, and raises
``IdealNerfError`` otherwise -- before the C call, and with no silent fallback.  The launch
happens on the tensors' own device and on that device's current stream, whatever the
process-wide current device is (``Network.to('cuda:1')`` without ``set_device(1)`` is legal,
as it is for the reference's torch ops).
"""
import contextlib
import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import IDN_PREC_F32, RAY_FLOATS, IdealNerfError, check

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


def _shape(t: Optional[torch.Tensor], name: str, *shape):
    """t must have exactly this shape (None = any size in that position)."""
    if t is None:
        return
    got = tuple(t.shape)
    if len(got) != len(shape) or any(s is not None and int(s) != g for s, g in zip(shape, got)):
        want = "[" + ", ".join("*" if s is None else str(int(s)) for s in shape) + "]"
        raise IdealNerfError(f"{name} must be {want}, got {list(got)}")


class _Launch:
    """One C call: every tensor argument on one GPU; that GPU current for the duration of the call
    (the library launches on the calling thread's current HIP device) and its current stream."""

    def __init__(self, *tensors, device=None):
        devs = {t.device for t in tensors if t is not None and t.is_cuda}
        if device is not None:
            d = torch.device(device)
            if d.type != "cuda":
                raise IdealNerfError(f"the HIP path renders on the GPU (got device {device}); there is no CPU fallback")
            devs.add(torch.device("cuda", torch.cuda.current_device() if d.index is None else d.index))
        if len(devs) > 1:
            raise IdealNerfError("all tensors of one call must live on the same GPU, got " +
                                 ", ".join(sorted(str(d) for d in devs)))
        self.device = devs.pop() if devs else None
        self._ctx = contextlib.nullcontext()

    def __enter__(self):
        if self.device is not None:
            self._ctx = torch.cuda.device(self.device)
            self._ctx.__enter__()
            self.stream = torch.cuda.current_stream(self.device).cuda_stream
        else:   # only CPU tensors: let _ptr raise the "must live on the GPU" error
            self.stream = None
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


def _precision(precision):
    if precision not in (_lib.IDN_PREC_F32, _lib.IDN_PREC_BF16X3, _lib.IDN_PREC_BF16, _lib.IDN_PREC_FP16X3, _lib.IDN_PREC_BF16X6):
        raise IdealNerfError(f"unknown precision code {precision}")
    return int(precision)


def _net_buffers(lib, packed, folded, precision, which=""):
    """The two per-network device buffers must have the sizes the kernels walk."""
    if packed is not None:
        _shape(packed, f"packed{which}", lib.idealnerf_packed_weight_floats(_precision(precision)))
    if folded is not None:
        _shape(folded, f"folded{which}", lib.idealnerf_folded_bias_floats())


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
    """`p` holds raw addresses: the caller (FaceNeRF.packed_weights) guarantees they live on `device`."""
    lib = _lib.load()
    with _Launch(device=device) as L:
        out = torch.empty(lib.idealnerf_packed_weight_floats(_precision(precision)), dtype=torch.float32, device=L.device)
        check(lib.idealnerf_pack_weights(C.byref(p), precision, out.data_ptr(), L.stream))
    return out


def fold_conditioning(p, aud, expr, latent, device) -> torch.Tensor:
    lib = _lib.load()
    _shape(aud, "aud", p.dim_aud)
    _shape(expr, "expr", p.dim_expr)
    _shape(latent, "latent", p.dim_latent)
    if aud is None and p.dim_aud:
        raise IdealNerfError(f"aud is None but the network was built with dim_aud={p.dim_aud}")
    with _Launch(aud, expr, latent, device=device) as L:
        out = torch.empty(lib.idealnerf_folded_bias_floats(), dtype=torch.float32, device=L.device)
        check(lib.idealnerf_fold_conditioning(C.byref(p), _ptr(aud, "aud"), _ptr(expr, "expr"), _ptr(latent, "latent"),
                                              out.data_ptr(), L.stream))
    return out


def facenerf_fwd(packed, folded, x, precision=IDN_PREC_F32) -> torch.Tensor:
    lib = _lib.load()
    _shape(x, "x", None, PTS_CH + VIEWS_CH)
    _net_buffers(lib, packed, folded, precision)
    with _Launch(packed, folded, x) as L:
        out = torch.empty((x.shape[0], 4), dtype=torch.float32, device=x.device)
        check(lib.idealnerf_facenerf_fwd(_ptr(packed, "packed"), _ptr(folded, "folded"), precision, _ptr(x, "x"),
                                         x.shape[0], out.data_ptr(), L.stream))
    return out


def query_rays_fwd(packed, folded, rays, z, precision=IDN_PREC_F32) -> torch.Tensor:
    lib = _lib.load()
    _shape(z, "z", None, None)
    n, S = z.shape
    _shape(rays, "rays", n, RAY_FLOATS)
    _net_buffers(lib, packed, folded, precision)
    with _Launch(packed, folded, rays, z) as L:
        raw = torch.empty((n, S, 4), dtype=torch.float32, device=z.device)
        check(lib.idealnerf_query_rays_fwd(_ptr(packed, "packed"), _ptr(folded, "folded"), precision, _ptr(rays, "rays"),
                                           _ptr(z, "z"), n, S, raw.data_ptr(), L.stream))
    return raw


def query_points_fwd(packed, folded, pts, viewdirs, precision=IDN_PREC_F32) -> torch.Tensor:
    lib = _lib.load()
    _shape(pts, "pts", None, None, 3)
    n, S, _ = pts.shape
    _shape(viewdirs, "viewdirs", n, 3)
    _net_buffers(lib, packed, folded, precision)
    with _Launch(packed, folded, pts, viewdirs) as L:
        raw = torch.empty((n, S, 4), dtype=torch.float32, device=pts.device)
        check(lib.idealnerf_query_points_fwd(_ptr(packed, "packed"), _ptr(folded, "folded"), precision, _ptr(pts, "pts"),
                                             _ptr(viewdirs, "viewdirs"), n, S, raw.data_ptr(), L.stream))
    return raw


def frame_rays(c2w, H, W, focal, near, far, row0=0, nrows=None, cx=None, cy=None, device="cuda") -> torch.Tensor:
    lib = _lib.load()
    nrows = H - row0 if nrows is None else nrows
    if not (0 <= row0 and 0 <= nrows and row0 + nrows <= H and W > 0):
        raise IdealNerfError(f"rows [{row0}, {row0 + nrows}) are outside a {H}x{W} frame")
    if tuple(c2w.shape[-2:]) not in ((3, 4), (4, 4)) or c2w.dim() != 2:
        raise IdealNerfError(f"c2w must be [3, 4] or [4, 4], got {list(c2w.shape)}")
    m = (C.c_float * 12)(*[float(v) for v in c2w[:3, :4].reshape(-1).tolist()])
    with _Launch(device=device) as L:
        out = torch.empty((nrows * W, RAY_FLOATS), dtype=torch.float32, device=L.device)
        check(lib.idealnerf_frame_rays(m, H, W, float(focal), -1.0 if cx is None else float(cx),
                                       -1.0 if cy is None else float(cy), float(near), float(far), row0, nrows,
                                       out.data_ptr(), L.stream))
    return out


def to8b(rgb, swap_rb=False, nonfinite_flag=None) -> torch.Tensor:
    """helper.py:154 on the device: [..., 3] fp32 -> [..., 3] uint8.  `nonfinite_flag` (int32[1] on
    the device) is OR-ed with 1 when the frame holds a NaN/Inf."""
    lib = _lib.load()
    if rgb.shape[-1] != 3:
        raise IdealNerfError(f"to8b expects [..., 3], got {tuple(rgb.shape)}")
    if nonfinite_flag is not None and (nonfinite_flag.dtype != torch.int32 or not nonfinite_flag.is_cuda
                                       or nonfinite_flag.numel() < 1):
        raise IdealNerfError("nonfinite_flag must be an int32 device tensor")
    with _Launch(rgb, nonfinite_flag) as L:
        out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
        check(lib.idealnerf_to8b(_ptr(rgb, "rgb"), rgb.numel() // 3, int(bool(swap_rb)), out.data_ptr(),
                                 nonfinite_flag.data_ptr() if nonfinite_flag is not None else None, L.stream))
    return out


def coarse_depths(rays, t_vals, t_rand=None, lindisp=False) -> torch.Tensor:
    lib = _lib.load()
    _shape(rays, "rays", None, RAY_FLOATS)
    _shape(t_vals, "t_vals", None)
    n, S = rays.shape[0], t_vals.shape[0]
    _shape(t_rand, "t_rand", n, S)
    with _Launch(rays, t_vals, t_rand) as L:
        z = torch.empty((n, S), dtype=torch.float32, device=rays.device)
        check(lib.idealnerf_coarse_depths(_ptr(rays, "rays"), _ptr(t_vals, "t_vals"), _ptr(t_rand, "t_rand"),
                                          int(bool(lindisp)), n, S, z.data_ptr(), L.stream))
    return z


def composite_fwd(raw, z, rays, bc_rgb, with_fg=False, with_weights=True, sigma_noise=None,
                  white_bkgd=False) -> Dict[str, torch.Tensor]:
    lib = _lib.load()
    _shape(z, "z", None, None)
    n, S = z.shape
    _shape(raw, "raw", n, S, 4)
    _shape(rays, "rays", n, RAY_FLOATS)
    _shape(bc_rgb, "bc_rgb", n, 3)
    _shape(sigma_noise, "sigma_noise", n, S)
    dev = z.device
    with _Launch(raw, z, rays, bc_rgb, sigma_noise) as L:
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        o = dict(rgb_map=new(n, 3), disp_map=new(n), acc_map=new(n), depth_map=new(n), last_weight=new(n))
        if with_weights:
            o["weights"] = new(n, S)
        if with_fg:
            o["rgb_fg"] = new(n, 3)
        co = _lib.CompositeOut(**{k: v.data_ptr() for k, v in o.items()})
        check(lib.idealnerf_composite_fwd(_ptr(raw, "raw"), _ptr(z, "z"), _ptr(rays, "rays"), _ptr(bc_rgb, "bc_rgb"),
                                          _ptr(sigma_noise, "sigma_noise"), int(bool(white_bkgd)), n, S, C.byref(co), L.stream))
    return o


def _u_shape(u, n, n_importance=None):
    """u: [Ni] shared by all rays (the deterministic linspace) or [n, Ni] per ray."""
    if u.dim() == 2:
        _shape(u, "u", n, n_importance)
        return 1
    _shape(u, "u", n_importance)
    return 0


def sample_pdf_fwd(z, weights, u, n_importance) -> Dict[str, torch.Tensor]:
    lib = _lib.load()
    _shape(z, "z", None, None)
    n, S = z.shape
    _shape(weights, "weights", n, S)
    per_ray = _u_shape(u, n, n_importance)
    dev = z.device
    with _Launch(z, weights, u) as L:
        o = dict(z_samples=torch.empty((n, n_importance), dtype=torch.float32, device=dev),
                 inds=torch.empty((n, n_importance), dtype=torch.int64, device=dev),
                 cdf=torch.empty((n, S - 1), dtype=torch.float32, device=dev),
                 z_fine=torch.empty((n, S + n_importance), dtype=torch.float32, device=dev),
                 z_std=torch.empty((n,), dtype=torch.float32, device=dev))
        check(lib.idealnerf_sample_pdf_fwd(_ptr(z, "z"), _ptr(weights, "weights"), _ptr(u, "u"), per_ray, n, S,
                                           n_importance, o["z_samples"].data_ptr(), o["inds"].data_ptr(),
                                           o["cdf"].data_ptr(), o["z_fine"].data_ptr(), o["z_std"].data_ptr(), L.stream))
    return o


def march_fwd(raw, z, rays, bc_rgb, u, n_importance, with_fg=False, with_weights=False, sigma_noise=None,
              white_bkgd=False) -> Dict[str, torch.Tensor]:
    """Coarse raw2outputs + sample_pdf + merge as one kernel (audio_exp_nerf.py:335-349): the union of
    composite_fwd's and sample_pdf_fwd's outputs; `weights` only when asked for."""
    lib = _lib.load()
    _shape(z, "z", None, None)
    n, S = z.shape
    _shape(raw, "raw", n, S, 4)
    _shape(rays, "rays", n, RAY_FLOATS)
    _shape(bc_rgb, "bc_rgb", n, 3)
    _shape(sigma_noise, "sigma_noise", n, S)
    per_ray = _u_shape(u, n, n_importance)
    dev = z.device
    with _Launch(raw, z, rays, bc_rgb, u, sigma_noise) as L:
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        o = dict(rgb_map=new(n, 3), disp_map=new(n), acc_map=new(n), depth_map=new(n), last_weight=new(n))
        if with_weights:
            o["weights"] = new(n, S)
        if with_fg:
            o["rgb_fg"] = new(n, 3)
        co = _lib.CompositeOut(**{k: v.data_ptr() for k, v in o.items()})
        o.update(z_samples=new(n, n_importance), inds=torch.empty((n, n_importance), dtype=torch.int64, device=dev),
                 cdf=new(n, S - 1), z_fine=new(n, S + n_importance), z_std=new(n))
        check(lib.idealnerf_march_fwd(_ptr(raw, "raw"), _ptr(z, "z"), _ptr(rays, "rays"), _ptr(bc_rgb, "bc_rgb"),
                                      _ptr(sigma_noise, "sigma_noise"), int(bool(white_bkgd)), _ptr(u, "u"), per_ray, n, S,
                                      n_importance, C.byref(co), o["z_samples"].data_ptr(), o["inds"].data_ptr(),
                                      o["cdf"].data_ptr(), o["z_fine"].data_ptr(), o["z_std"].data_ptr(), L.stream))
    return o


def sample_pdf_bins_fwd(bins, weights, u) -> Dict[str, torch.Tensor]:
    """helper.sample_pdf's own argument list (helper.py:269): bins [n, nb], weights [n, nb-1] (already
    the interior weights), u [Ni] or [n, Ni] -> z_samples, inds, cdf -- pdf, cdf and inversion all by
    the kernel that Network.render_rays runs."""
    lib = _lib.load()
    _shape(bins, "bins", None, None)
    n, nb = bins.shape
    _shape(weights, "weights", n, nb - 1)
    per_ray = _u_shape(u, n)
    ni = u.shape[-1]
    dev = bins.device
    with _Launch(bins, weights, u) as L:
        o = dict(z_samples=torch.empty((n, ni), dtype=torch.float32, device=dev),
                 inds=torch.empty((n, ni), dtype=torch.int64, device=dev),
                 cdf=torch.empty((n, nb), dtype=torch.float32, device=dev))
        check(lib.idealnerf_sample_pdf_bins_fwd(_ptr(bins, "bins"), _ptr(weights, "weights"), _ptr(u, "u"), per_ray, n,
                                                nb, ni, o["z_samples"].data_ptr(), o["inds"].data_ptr(),
                                                o["cdf"].data_ptr(), L.stream))
    return o


def invert_cdf(cdf, bins, u):
    lib = _lib.load()
    _shape(cdf, "cdf", None, None)
    n, nb = cdf.shape
    _shape(bins, "bins", n, nb)
    per_ray = _u_shape(u, n)
    ni = u.shape[-1]
    with _Launch(cdf, bins, u) as L:
        zs = torch.empty((n, ni), dtype=torch.float32, device=cdf.device)
        inds = torch.empty((n, ni), dtype=torch.int64, device=cdf.device)
        check(lib.idealnerf_invert_cdf(_ptr(cdf, "cdf"), _ptr(bins, "bins"), _ptr(u, "u"), per_ray, n, nb, ni,
                                       zs.data_ptr(), inds.data_ptr(), L.stream))
    return zs, inds


def dw_gemm(delta, acts, pipe=0, with_bias=True):
    """Test aid: one 256 x 256 weight-gradient product of the training step in isolation (idealnerf_dw_gemm):
    delta [rows, >= 256], acts [rows, >= 256] (row pitch = their width) -> dW [256, 256] = delta[:, :256]^T acts[:, :256]
    and db [256] = column sums of delta.  pipe 0: six bf16 piece products (what the step runs), 1: fp32 MFMA."""
    lib = _lib.load()
    _shape(delta, "delta", None, None)
    rows = delta.shape[0]
    _shape(acts, "acts", rows, None)
    with _Launch(delta, acts) as L:
        dW = torch.empty((256, 256), dtype=torch.float32, device=delta.device)
        db = torch.empty(256, dtype=torch.float32, device=delta.device) if with_bias else None
        nbytes = lib.idealnerf_dw_gemm_workspace_bytes()
        ws = _workspace(nbytes, delta.device, L.stream)
        check(lib.idealnerf_dw_gemm(_ptr(delta, "delta"), delta.shape[1], _ptr(acts, "acts"), acts.shape[1], rows, dW.data_ptr(),
                                    None if db is None else db.data_ptr(), int(pipe), ws.data_ptr(), ws.numel(), L.stream))
    return dW, db


AUDIO_NET_MAX_BWD_WINDOWS = 8


def _audio_params_struct(params, dim_aud):
    """params: encoder_conv.{0,2,4,6}.{weight,bias} then encoder_fc1.{0,2}.{weight,bias} (12 tensors, nn layout)."""
    shapes = [(32, 29, 3), (32,), (32, 32, 3), (32,), (64, 32, 3), (64,), (64, 64, 3), (64,), (64, 64), (64,), (dim_aud, 64), (dim_aud,)]
    if len(params) != 12:
        raise IdealNerfError(f"AudioNet has 12 parameter tensors, got {len(params)}")
    for i, (t, shp) in enumerate(zip(params, shapes)):
        _shape(t, f"AudioNet parameter {i}", *shp)
    p = _lib.AudioNetParams()
    for i in range(4):
        p.conv_w[i], p.conv_b[i] = _ptr(params[2 * i], "conv weight"), _ptr(params[2 * i + 1], "conv bias")
    for i in range(2):
        p.fc_w[i], p.fc_b[i] = _ptr(params[8 + 2 * i], "fc weight"), _ptr(params[9 + 2 * i], "fc bias")
    p.dim_aud = int(dim_aud)
    return p


def audio_net_fwd(params, windows, dim_aud, save=False):
    """AudioNet.forward (models/audio_net.py:43-69) on [n, 16, 29] windows as one kernel -> ([n, dim_aud], saved or None)."""
    lib = _lib.load()
    _shape(windows, "windows", None, 16, 29)
    n = windows.shape[0]
    with _Launch(windows, *params) as L:
        p = _audio_params_struct(params, dim_aud)
        out = torch.empty((n, dim_aud), dtype=torch.float32, device=windows.device)
        saved = torch.empty(lib.idealnerf_audio_net_saved_floats(n), dtype=torch.float32, device=windows.device) if save else None
        check(lib.idealnerf_audio_net_fwd(C.byref(p), _ptr(windows, "windows"), n, out.data_ptr(),
                                          None if saved is None else saved.data_ptr(), L.stream))
    return out, saved


def audio_net_bwd(params, windows, saved, d_out, dim_aud):
    """-> the 12 parameter gradients (summed over the n <= 8 windows), in the order of `params`."""
    lib = _lib.load()
    n = windows.shape[0]
    _shape(windows, "windows", None, 16, 29)
    _shape(d_out, "d_out", n, dim_aud)
    _shape(saved, "saved", lib.idealnerf_audio_net_saved_floats(n))
    with _Launch(windows, saved, d_out, *params) as L:
        p = _audio_params_struct(params, dim_aud)
        grads = [torch.empty_like(t) for t in params]
        g = _lib.AudioNetGrads()
        for i in range(4):
            g.conv_w[i], g.conv_b[i] = grads[2 * i].data_ptr(), grads[2 * i + 1].data_ptr()
        for i in range(2):
            g.fc_w[i], g.fc_b[i] = grads[8 + 2 * i].data_ptr(), grads[9 + 2 * i].data_ptr()
        check(lib.idealnerf_audio_net_bwd(C.byref(p), C.byref(g), _ptr(windows, "windows"), _ptr(saved, "saved"), _ptr(d_out, "d_out"),
                                          n, L.stream))
    return grads


_workspaces: Dict[tuple, torch.Tensor] = {}


def _workspace(nbytes: int, device, stream) -> torch.Tensor:
    """Scratch of one (device, stream): calls on one stream are ordered, so they can share it."""
    key = (str(device), int(stream))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        # zeros, once per allocation: the backward multiplies whole 256-column delta matrices of which it writes 129 columns
        # (csrc/train.hip: views_linears.0 + alpha_linear); what the other columns hold is never read back, but it must be
        # DEFINED data (zeros now, values our own kernels wrote later), not whatever the allocator hands out
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


_FUSED_NAMES = {"0": 0, "1": 1, "2": 2, "split": 2}


def _fused_code(fused, where) -> int:
    """False / 0 -> 0 (kernel sequence), True / 1 -> 1 (one kernel), "split" / 2 -> 2 (two kernels); anything else is named."""
    if isinstance(fused, str) and fused.strip().lower() in _FUSED_NAMES:
        return _FUSED_NAMES[fused.strip().lower()]
    if isinstance(fused, (bool, int)) and int(fused) in (0, 1, 2):
        return int(fused)
    raise IdealNerfError(f"{where} must be one of 0 / False (kernel sequence), 1 / True (one fused kernel), 2 / 'split' "
                         f"(two fused kernels); got {fused!r}")


# read once: the arrangement of the fp32 64 + 128 path where `fused` is not given
FUSED_MARCH_DEFAULT = _fused_code(os.environ.get("IDN_FUSED_MARCH", "0") or "0", "IDN_FUSED_MARCH")


def make_frame(c2w, H, W, focal, near, far, row0=0, nrows=None, cx=None, cy=None):
    """The camera of a full-frame render (idn_frame): rays of rows [row0, row0 + nrows) are derived on the device."""
    nrows = H - row0 if nrows is None else nrows
    if not (0 <= row0 and 0 <= nrows and row0 + nrows <= H and W > 0):
        raise IdealNerfError(f"rows [{row0}, {row0 + nrows}) are outside a {H}x{W} frame")
    if c2w.dim() != 2 or tuple(c2w.shape[-2:]) not in ((3, 4), (4, 4)):
        raise IdealNerfError(f"c2w must be [3, 4] or [4, 4], got {list(c2w.shape)}")
    f = _lib.Frame()
    f.c2w = (C.c_float * 12)(*[float(v) for v in c2w[:3, :4].reshape(-1).tolist()])
    f.H, f.W, f.focal = int(H), int(W), float(focal)
    f.cx, f.cy = -1.0 if cx is None else float(cx), -1.0 if cy is None else float(cy)
    f.near_, f.far_, f.row0, f.nrows = float(near), float(far), int(row0), int(nrows)
    f.rays_out = None
    return f


def philox_uniform(seed, which, row0, n_rows, n_cols, device) -> torch.Tensor:
    """Rows [row0, row0 + n_rows) of the table the in-kernel draws come from (`draws=` of render_rays_fwd) as a tensor:
    which = 0 the stratified offsets t_rand, 1 the importance draws u (include/idealnerf.h: idealnerf_philox_uniform)."""
    lib = _lib.load()
    with _Launch(device=device) as L:   # (raises for a CPU device: there is no CPU path)
        out = torch.empty((int(n_rows), int(n_cols)), dtype=torch.float32, device=L.device)
        check(lib.idealnerf_philox_uniform(int(seed) & (2 ** 64 - 1), int(which), int(row0), int(n_rows), int(n_cols), _ptr(out, "out"), L.stream))
    return out


def render_rays_fwd(rays, bc_rgb, packed_c, folded_c, packed_f, folded_f, t_vals, u, n_importance,
                    t_rand=None, with_fg=False, taps=False, precision=IDN_PREC_F32, precision_fine=None, lindisp=False,
                    white_bkgd=False, noise_coarse=None, noise_fine=None, fused=None, frame=None, draws=None) -> Dict[str, torch.Tensor]:
    """Network.render_rays forward (audio_exp_nerf.py:297-371) as one C call.  `precision_fine` (default: the
    same as `precision`) selects the fine network's arithmetic; packed_f must be packed for it.
    `fused`: the arrangement of the kernels (same results bit for bit).  False / 0: the kernel sequence
    (network, march, network, compositing) -- the default.  True / 1: the whole path as ONE kernel with a ray's samples, raw
    outputs, weights and cdf in LDS (csrc/render_fused.hip): 0-0.5 % faster on a full frame, six times the HBM bytes (it
    re-fetches a network's 2.4 MB weight stream into every L2 at each change of network).  "split" / 2: the same kernel as
    two launches (coarse network + march | fine network + compositing): half the sequence's bytes, 0.5 % slower.  1 and 2 are
    built for fp32, 64 + 128 samples, no density noise -- anything else raises.  None: IDN_FUSED_MARCH (0 / 1 / 2, read once)
    wherever the fused kernel applies, else the sequence.
    `frame` (ops.make_frame) with `rays=None`: full-frame mode (idealnerf_render_frame_fwd) -- the ray records of the frame's
    row band are derived on the device pass by pass; with `taps` they come back as `tap_rays`.
    `draws=(seed, ray0)` with `u=None, t_rand=None`: the perturb > 0 draws are made inside the kernels (rng_mode 1 of
    idn_render_args), ray r using row ray0 + r of the table `philox_uniform` writes out -- no [n, S] / [n, Ni] random tensors."""
    lib = _lib.load()
    if draws is not None and (u is not None or t_rand is not None):
        raise IdealNerfError("draws=(seed, ray0) replaces BOTH t_rand and u: pass them as None")
    _shape(t_vals, "t_vals", None)
    if frame is not None:
        if rays is not None:
            raise IdealNerfError("frame mode derives the rays from the camera: pass rays=None")
        n = frame.nrows * frame.W
    else:
        _shape(rays, "rays", None, RAY_FLOATS)
        n = rays.shape[0]
    S, Ni = t_vals.shape[0], int(n_importance)
    _shape(bc_rgb, "bc_rgb", n, 3)
    _shape(t_rand, "t_rand", n, S)
    _shape(noise_coarse, "noise_coarse", n, S)
    _shape(noise_fine, "noise_fine", n, S + Ni)
    _net_buffers(lib, packed_c, folded_c, precision, "_coarse")
    if Ni > 0:
        if (u is None and draws is None) or packed_f is None or folded_f is None:
            raise IdealNerfError("n_importance > 0 needs u (or draws=) and the fine network's packed / folded buffers")
        if u is not None:
            _u_shape(u, n, Ni)
        _net_buffers(lib, packed_f, folded_f, precision if precision_fine is None else precision_fine, "_fine")
    dev = bc_rgb.device
    with _Launch(rays, bc_rgb, packed_c, folded_c, packed_f, folded_f, t_vals, u, t_rand, noise_coarse, noise_fine) as L:
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
        a = _lib.RenderArgs()
        a.rays, a.bc_rgb, a.n_rays = _ptr(rays, "rays"), _ptr(bc_rgb, "bc_rgb"), n
        a.n_samples, a.n_importance, a.precision = S, Ni, _precision(precision)
        a.precision_fine_plus1 = 0 if precision_fine is None else _precision(precision_fine) + 1
        a.packed_coarse, a.folded_coarse = _ptr(packed_c, "packed_coarse"), _ptr(folded_c, "folded_coarse")
        a.packed_fine, a.folded_fine = _ptr(packed_f, "packed_fine"), _ptr(folded_f, "folded_fine")
        a.t_vals, a.t_rand = _ptr(t_vals, "t_vals"), _ptr(t_rand, "t_rand")
        a.u = _ptr(u, "u") if u is not None else None
        a.u_per_ray = 1 if (u is not None and u.dim() == 2) else 0
        a.lindisp, a.white_bkgd = int(bool(lindisp)), int(bool(white_bkgd))
        a.noise_coarse, a.noise_fine = _ptr(noise_coarse, "noise_coarse"), _ptr(noise_fine, "noise_fine")
        if draws is not None:
            a.rng_mode, a.rng_seed, a.rng_ray0 = 1, int(draws[0]) & (2 ** 64 - 1), int(draws[1])
        if fused is None:
            applies = (S == 64 and Ni == 128 and noise_coarse is None and noise_fine is None and
                       _precision(precision) == IDN_PREC_F32 and (precision_fine is None or _precision(precision_fine) == IDN_PREC_F32))
            fused = FUSED_MARCH_DEFAULT if applies else 0
        a.fused_march = _fused_code(fused, "render_rays_fwd(fused=)")
        for k, v in out.items():
            setattr(a, k, v.data_ptr())
        if frame is None:
            nbytes = lib.idealnerf_render_workspace_bytes(n, S, Ni)
            ws = _workspace(nbytes, dev, L.stream or 0)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            check(lib.idealnerf_render_rays_fwd(C.byref(a), L.stream))
        else:
            tap_rays = new(n, RAY_FLOATS) if taps else None
            frame.rays_out = None if tap_rays is None else tap_rays.data_ptr()
            nbytes = lib.idealnerf_render_frame_workspace_bytes(n, S, Ni)
            ws = _workspace(nbytes, dev, L.stream or 0)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            check(lib.idealnerf_render_frame_fwd(C.byref(a), C.byref(frame), L.stream))
            frame.rays_out = None
            if tap_rays is not None:
                out["tap_rays"] = tap_rays
    return out
