"""CPU restatement (PyTorch-CPU, fp32) of IDEAL-NeRF's per-ray hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Written from the math of the reference, not from its code.  Every function
names the reference lines (relative to the upstream repo root) whose behaviour
it restates.  The op *sequence* follows the reference's eager-op sequence where
that decides fp32 rounding (one rounding per eager op, ``torch.cumprod`` /
``torch.cumsum`` / ``torch.sum`` as PyTorch-CPU implements them), because the
importance-sampling indices are a discontinuous function of those roundings.

Weights are plain dicts keyed by the reference's ``state_dict`` key names
(``pts_linears.0.weight`` ...), ``nn.Linear`` layout ``[out, in]``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

__all__ = [
    "positional_encoding", "pe_out_dim", "camera_rays", "ray_records",
    "facenerf_dims", "facenerf_param_shapes", "xavier_facenerf_params",
    "facenerf_forward", "facenerf_forward_bf16_emulated", "composite", "importance_cdf", "invert_cdf",
    "sample_importance", "coarse_depths", "render_rays", "render_frame",
    "torso_signal", "pose_to_euler_trans", "head_torso_composite", "train_loss", "to_f64", "fp32_noise_floor",
    "mse_to_psnr", "synthetic_frame",
]

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# a2  positional encoding            NeRFs/HeadNeRF/helper.py:174-224
# --------------------------------------------------------------------------
def pe_out_dim(n_freqs: int, in_dims: int = 3) -> int:
    return in_dims * (1 + 2 * n_freqs)


def positional_encoding(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)].

    Output order is band-major, then sin/cos, then the input axis
    (helper.py:183-201: one lambda per (freq, fn), each applied to the whole
    last axis; helper.py:203-204: concatenated on the last axis).  The bands
    are 2**linspace(0, L-1, L) (helper.py:190-191): exact powers of two.
    """
    feats = [x]
    bands = 2.0 ** torch.linspace(0.0, float(n_freqs - 1), steps=n_freqs)
    for f in bands:
        xf = x * f
        feats.append(torch.sin(xf))
        feats.append(torch.cos(xf))
    return torch.cat(feats, dim=-1)


# --------------------------------------------------------------------------
# a1  pinhole rays + ray records     helper.py:228-243, audio_exp_nerf.py:396-427
# --------------------------------------------------------------------------
def camera_rays(H: int, W: int, focal: float, c2w: torch.Tensor,
                cx: Optional[float] = None, cy: Optional[float] = None):
    """rays_o, rays_d of shape [H, W, 3] for a pinhole camera looking down -z.

    Pixel (row j, column i) has camera-frame direction ((i-cx)/f, -(j-cy)/f, -1)
    (helper.py:231-237; default principal point W/2, H/2), rotated by
    c2w[:3,:3] as sum_k dir_k * R[r, k] (helper.py:240), origin c2w[:3, 3]
    (helper.py:242).
    """
    cols = torch.linspace(0, W - 1, W)
    rows = torch.linspace(0, H - 1, H)
    i = cols[None, :].expand(H, W)
    j = rows[:, None].expand(H, W)
    cx = W * 0.5 if cx is None else cx
    cy = H * 0.5 if cy is None else cy
    dirs = torch.stack([(i - cx) / focal, -(j - cy) / focal, -torch.ones_like(i)], dim=-1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], dim=-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def ray_records(rays_o: torch.Tensor, rays_d: torch.Tensor, near: float, far: float) -> torch.Tensor:
    """[n, 11] records (o, d, near, far, d/|d|)   (audio_exp_nerf.py:407-427)."""
    d = rays_d.reshape(-1, 3).float()
    o = rays_o.reshape(-1, 3).float()
    view = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    view = view.reshape(-1, 3).float()
    nr = near * torch.ones_like(d[..., :1])
    fr = far * torch.ones_like(d[..., :1])
    return torch.cat([o, d, nr, fr, view], dim=-1)


# --------------------------------------------------------------------------
# a5  FaceNeRF                        models/face_nerf.py:8-80
# --------------------------------------------------------------------------
def facenerf_dims(dim_aud=64, dim_expr=76, dim_latent=32, input_ch=63, input_ch_views=27, W=256, D=8, skips=(4,)):
    return dict(dim_aud=dim_aud, dim_expr=dim_expr, dim_latent=dim_latent, input_ch=input_ch,
                input_ch_views=input_ch_views, W=W, D=D, skips=tuple(skips))


def facenerf_param_shapes(dims) -> Dict[str, tuple]:
    """state_dict key -> shape   (models/face_nerf.py:27-37)."""
    W, D = dims["W"], dims["D"]
    c_all = dims["input_ch"] + dims["dim_aud"] + dims["dim_expr"] + dims["dim_latent"]
    shapes = {}
    for i in range(D):
        if i == 0:
            fan_in = c_all
        elif (i - 1) in dims["skips"]:
            fan_in = W + c_all
        else:
            fan_in = W
        shapes[f"pts_linears.{i}.weight"] = (W, fan_in)
        shapes[f"pts_linears.{i}.bias"] = (W,)
    shapes["views_linears.0.weight"] = (W // 2, dims["input_ch_views"] + W + dims["dim_expr"])
    shapes["views_linears.0.bias"] = (W // 2,)
    for i in range(1, D // 4 + 1):
        shapes[f"views_linears.{i}.weight"] = (W // 2, W // 2)
        shapes[f"views_linears.{i}.bias"] = (W // 2,)
    shapes["feature_linear.weight"] = (W, W)   # built but never used (face_nerf.py:34 vs :66)
    shapes["feature_linear.bias"] = (W,)
    shapes["alpha_linear.weight"] = (1, W)
    shapes["alpha_linear.bias"] = (1,)
    shapes["rgb_linear.weight"] = (3, W // 2)
    shapes["rgb_linear.bias"] = (3,)
    return shapes


def xavier_facenerf_params(seed: int, dims, gain: float = 1.0) -> Params:
    """Xavier-uniform weights + bias 0.01 (audio_exp_nerf.py:442-448), drawn from
    numpy.random.RandomState(seed) in state_dict key order so that the golden
    generator and the tests can rebuild identical weights without storing them."""
    rs = np.random.RandomState(seed)
    out = {}
    for k, shp in facenerf_param_shapes(dims).items():
        if k.endswith(".weight"):
            bound = gain * math.sqrt(6.0 / (shp[0] + shp[1]))
            out[k] = torch.from_numpy(rs.uniform(-bound, bound, size=shp).astype(np.float32))
        else:
            out[k] = torch.full(shp, 0.01, dtype=torch.float32)
    return out


def _linear(p: Params, name: str, h: torch.Tensor) -> torch.Tensor:
    return torch.addmm(p[name + ".bias"], h, p[name + ".weight"].t())


def facenerf_forward(p: Params, x: torch.Tensor, aud: Optional[torch.Tensor],
                     expr: Optional[torch.Tensor] = None, latent: Optional[torch.Tensor] = None,
                     dims=None) -> torch.Tensor:
    """[N, input_ch+input_ch_views] -> [N, 4] = (rgb_raw(3), sigma_raw(1)).

    initial = [gamma(x) | aud | expr/3 | latent] with the three conditioning
    vectors shared by all rows (face_nerf.py:41-55; expr is scaled as
    ``expr * 1 / 3``, i.e. an fp32 divide, :49).  Eight ReLU layers with
    ``[initial | h]`` re-injected after layer index 4 (:57-62); sigma from the
    trunk output (:65); colour branch on [h | gamma(dir) | expr/3] (:67-73);
    ``feature_linear`` is NOT applied (:66).
    """
    dims = dims or facenerf_dims()
    n = x.shape[0]
    pts, views = torch.split(x, [dims["input_ch"], dims["input_ch_views"]], dim=-1)
    parts = [pts]
    if aud is not None:
        parts.append(aud[None, :].expand(n, -1))
    expr3 = None
    if expr is not None:
        expr3 = (expr * 1 / 3)[None, :].expand(n, -1)
        parts.append(expr3)
    if latent is not None:
        parts.append(latent[None, :].expand(n, -1))
    initial = torch.cat(parts, dim=-1)
    h = initial
    for i in range(dims["D"]):
        h = torch.relu(_linear(p, f"pts_linears.{i}", h))
        if i in dims["skips"]:
            h = torch.cat([initial, h], dim=-1)
    sigma = _linear(p, "alpha_linear", h)
    hv = [h, views]
    if expr3 is not None:
        hv.append(expr3)
    h = torch.cat(hv, dim=-1)
    for i in range(dims["D"] // 4 + 1):
        h = torch.relu(_linear(p, f"views_linears.{i}", h))
    rgb = _linear(p, "rgb_linear", h)
    return torch.cat([rgb, sigma], dim=-1)


def facenerf_forward_bf16_emulated(p: Params, x: torch.Tensor, aud: Optional[torch.Tensor],
                                   expr: Optional[torch.Tensor] = None, latent: Optional[torch.Tensor] = None,
                                   dims=None) -> torch.Tensor:
    """`facenerf_forward` (models/face_nerf.py:40-80) in the arithmetic BASELINE configs[4] allows ("bf16 MFMA MLP"):
    the rounding model of the product's plain-bf16 kernel, stated from the math so that the kernel can be held to it.

    * every weight that multiplies a PER-POINT input is rounded to bf16 once (round to nearest even);
    * every per-point layer input -- the 63 + 27 encoding features, each hidden activation after its ReLU -- is rounded
      to bf16; products are exact and summed without further rounding (here: float64; the kernel: fp32 MFMA accumulate,
      ~1e-7 apart);
    * the per-FRAME conditioning columns (aud | expr/3 | latent of pts_linears.0 / .5, expr/3 of views_linears.0;
      face_nerf.py:45-55,61,68-70) multiply constants: they are folded into the bias in fp32 with unrounded weights, as
      the product does once per frame;
    * sigma (alpha_linear) and rgb (rgb_linear) are bf16 products of the bf16 activations like any other layer; neither
      is ReLU'd; biases stay fp32.

    What this cannot pin to better than ~1e-4 on a rare point: an activation that lies within the accumulation noise of
    a bf16 rounding boundary rounds the other way in the kernel and moves by one bf16 ulp (2^-8 of its value)."""
    dims = dims or facenerf_dims()
    bf = lambda t: t.to(torch.float32).to(torch.bfloat16).to(torch.float64)
    d64 = lambda t: t.to(torch.float64)
    ci, cv = dims["input_ch"], dims["input_ch_views"]
    pts, views = torch.split(x, [ci, cv], dim=-1)
    cond = [t for t in (aud, None if expr is None else expr * 1 / 3, latent) if t is not None]
    cond = d64(torch.cat(cond)) if cond else None
    nc = 0 if cond is None else cond.numel()
    expr3 = None if expr is None else d64(expr * 1 / 3)

    def layer(name, h_cols, cond_cols, cvec, inputs):
        """bias' + sum_i bf16(inputs_i) @ bf16(W[:, cols_i]).T ; bias' = bias + W[:, cond_cols] @ cvec (fp32 fold)"""
        W, b = d64(p[name + ".weight"]), d64(p[name + ".bias"])
        if cvec is not None and cvec.numel():
            b = (b + W[:, cond_cols[0]:cond_cols[1]] @ cvec).to(torch.float32).to(torch.float64)
        out = b[None, :]
        for (c0, c1), h in zip(h_cols, inputs):
            out = out + bf(h) @ bf(p[name + ".weight"][:, c0:c1]).t()
        return out.to(torch.float32)

    c_all = ci + nc
    h = torch.relu(layer("pts_linears.0", [(0, ci)], (ci, c_all), cond, [pts]))
    for i in range(1, dims["D"]):
        if (i - 1) in dims["skips"]:
            h = torch.relu(layer(f"pts_linears.{i}", [(0, ci), (c_all, c_all + dims["W"])], (ci, c_all), cond, [pts, h]))
        else:
            h = torch.relu(layer(f"pts_linears.{i}", [(0, dims["W"])], None, None, [h]))
    sigma = layer("alpha_linear", [(0, dims["W"])], None, None, [h])
    Wd = dims["W"]
    ne = 0 if expr3 is None else expr3.numel()
    h = torch.relu(layer("views_linears.0", [(0, Wd), (Wd, Wd + cv)], (Wd + cv, Wd + cv + ne), expr3, [h, views]))
    for i in range(1, dims["D"] // 4 + 1):
        h = torch.relu(layer(f"views_linears.{i}", [(0, Wd // 2)], None, None, [h]))
    rgb = layer("rgb_linear", [(0, Wd // 2)], None, None, [h])
    return torch.cat([rgb, sigma], dim=-1)


# --------------------------------------------------------------------------
# a6  alpha compositing               NeRFs/HeadNeRF/train/baseline.py:325-375
#     (+ rgb_map_fg variant           NeRFs/TorsoNeRF/run_nerf.py:715-766)
# --------------------------------------------------------------------------
def composite(raw: torch.Tensor, z: torch.Tensor, rays_d: torch.Tensor, bc_rgb: torch.Tensor,
              with_fg: bool = False, sigma_noise: Optional[torch.Tensor] = None, white_bkgd: bool = False):
    """raw[n,S,4], z[n,S], d[n,3], bc[n,3] -> rgb_map, disp, acc, weights, depth (, rgb_fg).

    dists = [dz, 1e10] * |d| (baseline.py:345-349); colours sigmoid(raw_rgb) with
    the LAST sample's colour replaced by the background pixel (:351-352);
    alpha = 1 - exp(-(relu(sigma)+1e-6) * dists) (:340-342,:363);
    T_s = prod_{t<s}(1 - alpha_t + 1e-10) (:365-367); w = alpha*T;
    rgb = sum w c (:368); depth = sum w z (:370); disp = 1/max(1e-10, depth/sum w)
    (:371); acc = sum w (:372).  Torso variant: rgb_fg = sum_{s<S-1} w c
    (TorsoNeRF/run_nerf.py:757).  ``sigma_noise`` [n,S] is added to the density before its
    ReLU (the caller draws it: randn * raw_noise_std, baseline.py:353-361); ``white_bkgd`` adds
    1 - acc to the colour (:372-373).
    """
    n = z.shape[0]
    dz = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dz, torch.full((n, 1), 1e10, dtype=z.dtype)], dim=-1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    rgb = torch.cat([rgb[:, :-1, :], bc_rgb[:, None, :]], dim=1)
    sigma = raw[..., 3] if sigma_noise is None else raw[..., 3] + sigma_noise
    alpha = 1.0 - torch.exp(-(torch.relu(sigma) + 1e-6) * dists)
    trans = torch.cumprod(torch.cat([torch.ones((n, 1)), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, dim=-2)
    depth = torch.sum(weights * z, dim=-1)
    wsum = torch.sum(weights, dim=-1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / wsum)
    acc = torch.sum(weights, dim=-1)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc[..., None])
    if with_fg:
        rgb_fg = torch.sum(weights[:, :-1, None] * rgb[:, :-1, :], dim=-2)
        return rgb_map, disp, acc, weights, depth, rgb_fg
    return rgb_map, disp, acc, weights, depth


# --------------------------------------------------------------------------
# a7  hierarchical sampling           NeRFs/HeadNeRF/helper.py:269-313
# --------------------------------------------------------------------------
def importance_cdf(weights_inner: torch.Tensor) -> torch.Tensor:
    """w[n, S-2] -> cdf[n, S-1] = [0, cumsum((w+1e-5)/sum(w+1e-5))]  (helper.py:271-275)."""
    w = weights_inner + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    return torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)


def invert_cdf(cdf: torch.Tensor, bins: torch.Tensor, u: torch.Tensor):
    """The bit-exact boundary: identical (cdf, bins, u) in => identical inds out.

    inds = #{k : cdf[k] <= u} (searchsorted right=True, helper.py:297), int64;
    below = max(0, inds-1), above = min(len-1, inds) (:298-299);
    denom = cdf[above]-cdf[below], replaced by 1 where < 1e-5 (:307-308);
    z = bins[below] + (u-cdf[below])/denom * (bins[above]-bins[below]) (:309-310).
    Returns (z_samples fp32, inds int64).
    """
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b = torch.gather(cdf, -1, below)
    cdf_a = torch.gather(cdf, -1, above)
    bin_b = torch.gather(bins, -1, below)
    bin_a = torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b), inds


def sample_importance(bins: torch.Tensor, weights_inner: torch.Tensor, n_importance: int,
                      det: bool = True, u: Optional[torch.Tensor] = None):
    """bins[n,S-1], w[n,S-2] -> (z_samples[n,Ni], inds[n,Ni] int64, cdf[n,S-1], u[n,Ni]).

    det: u = linspace(0,1,Ni) for every ray (helper.py:279-281); otherwise the
    caller supplies ``u`` (the reference draws torch.rand, :283, or numpy seed 0
    under its ``pytest`` flag, :286-293).
    """
    cdf = importance_cdf(weights_inner)
    if u is None:
        if not det:
            raise ValueError("non-deterministic sampling needs an explicit u buffer")
        u = torch.linspace(0.0, 1.0, steps=n_importance)
    if u.dim() == 1:
        u = u.expand(list(cdf.shape[:-1]) + [n_importance])
    u = u.contiguous()
    z, inds = invert_cdf(cdf, bins, u)
    return z, inds, cdf, u


# --------------------------------------------------------------------------
# a3/a4/a8/a9  render_rays            NeRFs/HeadNeRF/train/audio_exp_nerf.py:297-394
# --------------------------------------------------------------------------
def coarse_depths(near: torch.Tensor, far: torch.Tensor, n_samples: int,
                  t_rand: Optional[torch.Tensor] = None, lindisp: bool = False) -> torch.Tensor:
    """near,far [n,1] -> z[n,S]: z = near(1-t)+far t, t = linspace(0,1,S), or with
    ``lindisp`` linear in inverse depth, z = 1/((1/near)(1-t) + (1/far) t)
    (audio_exp_nerf.py:306-312); if ``t_rand`` [n,S] is given, stratified
    jitter z = lower + (upper-lower)*t_rand with t_rand[:, -1] forced to 1
    (:314-330)."""
    t = torch.linspace(0.0, 1.0, steps=n_samples, dtype=near.dtype)
    if lindisp:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    else:
        z = near * (1.0 - t) + far * t
    z = z.expand(near.shape[0], n_samples)
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], dim=-1)
        lower = torch.cat([z[..., :1], mids], dim=-1)
        t_rand = t_rand.clone()
        t_rand[..., -1] = 1.0
        z = lower + (upper - lower) * t_rand
    return z


def _query(p: Params, pts: torch.Tensor, viewdirs: torch.Tensor, aud, expr, latent, dims,
           netchunk: int = 1 << 16) -> torch.Tensor:
    """pts[n,S,3] -> raw[n,S,4]: PE(10) of points, PE(4) of the ray's unit
    direction repeated per sample, MLP in netchunk slices (audio_exp_nerf.py:376-394)."""
    n, s, _ = pts.shape
    flat = pts.reshape(-1, 3)
    emb = positional_encoding(flat, 10)
    dirs = viewdirs[:, None, :].expand(n, s, 3).reshape(-1, 3)
    emb = torch.cat([emb, positional_encoding(dirs, 4)], dim=-1)
    outs = [facenerf_forward(p, emb[i:i + netchunk], aud, expr, latent, dims)
            for i in range(0, emb.shape[0], netchunk)]
    return torch.cat(outs, dim=0).reshape(n, s, 4)


def render_rays(rays: torch.Tensor, bc_rgb: torch.Tensor, coarse: Params, fine: Optional[Params],
                aud, expr, latent, n_samples: int = 64, n_importance: int = 128,
                dims=None, t_rand: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None,
                with_fg: bool = False, taps: bool = False, lindisp: bool = False, white_bkgd: bool = False,
                noise_coarse: Optional[torch.Tensor] = None,
                noise_fine: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """rays[n,11], bc_rgb[n,3] -> dict with the reference's keys
    (audio_exp_nerf.py:297-371): rgb_map, disp_map, acc_map, and when
    n_importance>0: rgb0, disp0, acc0, z_std (population std of the importance
    depths, :363) and last_weight (:364).  ``t_rand is None and u is None`` is
    the reference's perturb=0 mode (deterministic u).  ``with_fg`` adds the torso
    variant's rgb_map_fg / rgb_map_fg0 / last_weight0 (train_torso.py:326-345).
    """
    dims = dims or facenerf_dims()
    o, d = rays[:, 0:3], rays[:, 3:6]
    view = rays[:, -3:]
    near, far = rays[:, 6:7], rays[:, 7:8]
    z = coarse_depths(near, far, n_samples, t_rand, lindisp=lindisp)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    raw = _query(coarse, pts, view, aud, expr, latent, dims)
    comp = composite(raw, z, d, bc_rgb, with_fg=with_fg, sigma_noise=noise_coarse, white_bkgd=white_bkgd)
    rgb_map, disp, acc, weights, depth = comp[:5]
    out = {}
    tap = {}
    if taps:
        tap.update(z_coarse=z, raw_coarse=raw, weights_coarse=weights)
    if n_importance > 0:
        out["rgb0"], out["disp0"], out["acc0"] = rgb_map, disp, acc
        if with_fg:
            out["rgb_map_fg0"] = comp[5]
            out["last_weight0"] = weights[..., -1]
        z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
        z_s, inds, cdf, u_used = sample_importance(z_mid, weights[..., 1:-1], n_importance,
                                                   det=(u is None), u=u)
        z_s = z_s.detach()
        z, _ = torch.sort(torch.cat([z, z_s], dim=-1), dim=-1)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        raw = _query(fine, pts, view, aud, expr, latent, dims)
        comp = composite(raw, z, d, bc_rgb, with_fg=with_fg, sigma_noise=noise_fine, white_bkgd=white_bkgd)
        rgb_map, disp, acc, weights, depth = comp[:5]
        out["z_std"] = torch.std(z_s, dim=-1, unbiased=False)
        out["last_weight"] = weights[..., -1]
        if taps:
            tap.update(cdf=cdf, inds=inds, z_samples=z_s, z_fine=z, raw_fine=raw, weights_fine=weights)
    out["rgb_map"], out["disp_map"], out["acc_map"] = rgb_map, disp, acc
    if with_fg:
        out["rgb_map_fg"] = comp[5]
    out.update({"tap_" + k: v for k, v in tap.items()})
    return out


def render_frame(H: int, W: int, focal: float, c2w: torch.Tensor, near: float, far: float,
                 bc_img: torch.Tensor, coarse: Params, fine: Params, aud, expr, latent,
                 n_samples=64, n_importance=128, chunk=8192, dims=None, rows=None, **kw):
    """Full-frame path (audio_exp_nerf.py:396-446): rays from the pose with the
    default principal point, background flattened row-major, ``chunk`` rays per
    render_rays call (batchify_rays :281-295), outputs reshaped to [H, W, ...].
    ``rows=(r0, r1)`` renders only that row band (the multi-GPU partition)."""
    ro, rd = camera_rays(H, W, focal, c2w)
    if rows is not None:
        ro, rd = ro[rows[0]:rows[1]], rd[rows[0]:rows[1]]
        bc_img = bc_img[rows[0]:rows[1]]
    sh = rd.shape
    rays = ray_records(ro, rd, near, far)
    bc = bc_img.reshape(-1, 3)
    acc: Dict[str, list] = {}
    for i in range(0, rays.shape[0], chunk):
        r = render_rays(rays[i:i + chunk], bc[i:i + chunk], coarse, fine, aud, expr, latent,
                        n_samples, n_importance, dims, **kw)
        for k, v in r.items():
            acc.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0).reshape(list(sh[:-1]) + list(v[0].shape[1:])) for k, v in acc.items()}


# --------------------------------------------------------------------------
# a11  head + torso composite         NeRFs/TorsoNeRF/train_torso.py:237-271
# --------------------------------------------------------------------------
def pose_to_euler_trans(poses: torch.Tensor) -> torch.Tensor:
    """[b,3or4,4] -> [b,6] = (euler(3), translation(3))  (run_nerf_helpers.py:26-47):
    e2 = atan2(R00, -R01), e1 = asin(-R02), e0 = atan2(R22, R12)."""
    R = poses[:, :3, :3]
    e = torch.stack([torch.atan2(R[:, 2, 2], R[:, 1, 2]),
                     torch.asin(-R[:, 0, 2]),
                     torch.atan2(R[:, 0, 0], -R[:, 0, 1])], dim=1)
    return torch.cat([e, poses[:, :3, 3]], dim=1)


def torso_signal(aud_feature: torch.Tensor, pose: torch.Tensor, dim_aud_body: int = 64,
                 n_freqs: int = 3) -> torch.Tensor:
    """[aud[:dim_aud_body] | PE_3(euler) | PE_3(trans)]   (train_torso.py:238-240)."""
    et = pose_to_euler_trans(pose[None])
    emb = torch.cat([positional_encoding(et[:, :3], n_freqs), positional_encoding(et[:, 3:], n_freqs)], dim=1)
    return torch.cat([aud_feature[..., :dim_aud_body], emb.squeeze(0)], dim=-1)


def head_torso_composite(head: Dict[str, torch.Tensor], torso: Dict[str, torch.Tensor]):
    """rgb_com = rgb_head * w_last,torso + rgb_fg,torso; same for the coarse pair
    (train_torso.py:269-270)."""
    rgb_com = head["rgb_map"] * torso["last_weight"][..., None] + torso["rgb_map_fg"]
    rgb_com0 = head["rgb0"] * torso["last_weight0"][..., None] + torso["rgb_map_fg0"]
    return rgb_com, rgb_com0


# --------------------------------------------------------------------------
# a12  train-step loss                audio_exp_nerf.py:534-548, helper.py:148-151
# --------------------------------------------------------------------------
def mse_to_psnr(mse: torch.Tensor) -> torch.Tensor:
    return -10.0 * torch.log(mse) / torch.log(torch.tensor([10.0]))


def train_loss(out: Dict[str, torch.Tensor], target: torch.Tensor, latent: torch.Tensor,
               lc_weight: float = 0.0005):
    """loss = mse(rgb, tgt) + mse(rgb0, tgt) + 10 * lc_weight * ||latent||_2
    (audio_exp_nerf.py:540-548).  Returns (loss, img_loss)."""
    img_loss = torch.nn.functional.mse_loss(out["rgb_map"], target)
    loss = img_loss
    if "rgb0" in out:
        loss = loss + torch.nn.functional.mse_loss(out["rgb0"], target)
    loss = loss + 10.0 * (torch.norm(latent) * lc_weight)
    return loss, img_loss


# --------------------------------------------------------------------------
# synthetic workload -- shared by tests and bench.py
# --------------------------------------------------------------------------
def synthetic_frame(H: int = 512, W: int = 512, seed: int = 0, dims=None):
    """Seeded synthetic inputs of the shape BASELINE.json names: pose = small
    seeded rotation of [I | (0,0,0.877)], May near/far, uniform background,
    Gaussian audio/expression latents, ones latent code (audio_exp_nerf.py:482)."""
    dims = dims or facenerf_dims()
    rs = np.random.RandomState(seed)
    ang = rs.uniform(-0.08, 0.08, size=3)
    cx_, sx = math.cos(ang[0]), math.sin(ang[0])
    cy_, sy = math.cos(ang[1]), math.sin(ang[1])
    cz, sz = math.cos(ang[2]), math.sin(ang[2])
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    c2w = np.concatenate([Rz @ Ry @ Rx, np.array([[0.0], [0.0], [0.877]])], axis=1).astype(np.float32)
    bc = np.random.RandomState(seed + 1).uniform(0, 1, size=(H, W, 3)).astype(np.float32)
    rs2 = np.random.RandomState(seed + 100)
    aud = rs2.standard_normal(dims["dim_aud"]).astype(np.float32) if dims["dim_aud"] else None
    expr = rs2.standard_normal(dims["dim_expr"]).astype(np.float32) if dims["dim_expr"] else None
    latent = np.ones(dims["dim_latent"], dtype=np.float32) if dims["dim_latent"] else None
    t = lambda a: None if a is None else torch.from_numpy(a)
    return dict(H=H, W=W, focal=1200.0 * W / 450.0, c2w=torch.from_numpy(c2w),
                near=0.5772005200386048, far=1.1772005200386046, bc=torch.from_numpy(bc),
                aud=t(aud), expr=t(expr), latent=t(latent))


# --------------------------------------------------------------------------
# How far is the fp32 formula itself from exact arithmetic on a given scene?
# --------------------------------------------------------------------------
def to_f64(x):
    """Tensors (also inside dicts / lists / tuples) as float64; everything else unchanged."""
    if torch.is_tensor(x):
        return x.double() if x.is_floating_point() else x
    if isinstance(x, dict):
        return {k: to_f64(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to_f64(v) for v in x)
    return x


def fp32_noise_floor(ref32: torch.Tensor, ref64: torch.Tensor) -> float:
    """max |ref32 - ref64| / max |ref64|: the distance of the reference's OWN fp32 evaluation from the same
    formulas in fp64.  The importance sampling divides by bin masses floored at 1e-5, so where the density is
    sharp a last-ulp difference in a coarse weight relocates fine samples and moves a pixel by ~1e-4: on such
    scenes two correct fp32 implementations (CPU BLAS vs fp32 MFMA, or the same CPU code on two hosts) differ
    by about this much, and neither can be asked to match the other more closely."""
    a, b = ref32.detach().double(), ref64.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
