"""Training path: ``Network.render_rays`` with gradients
(NeRFs/HeadNeRF/train/audio_exp_nerf.py:534-552).

One ``torch.autograd.Function`` per render: forward runs the HIP stages with the
activation-saving MLP variant, backward is two C calls (fine pass, coarse pass), each
compositing-backward + 11 layers of MFMA GEMMs + the conditioning fold.  Gradients reach
every FaceNeRF parameter of both networks, ``aud_para`` (and through it the audio nets,
which stay ordinary autograd modules) and ``latent_code``.  Sampled depths are detached
exactly as upstream (:345); ``expr`` is data.
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import IDN_PREC_F32, check
from .helper import linspace01

PARAM_KEYS = ([f"pts_linears.{i}.{k}" for i in range(8) for k in ("weight", "bias")] +
              [f"views_linears.{i}.{k}" for i in range(3) for k in ("weight", "bias")] +
              ["alpha_linear.weight", "alpha_linear.bias", "rgb_linear.weight", "rgb_linear.bias"])

def _grads_struct(grads):
    g = _lib.FaceNerfGrads()
    for i in range(8):
        g.pts_w[i] = grads[f"pts_linears.{i}.weight"].data_ptr()
        g.pts_b[i] = grads[f"pts_linears.{i}.bias"].data_ptr()
    for i in range(3):
        g.views_w[i] = grads[f"views_linears.{i}.weight"].data_ptr()
        g.views_b[i] = grads[f"views_linears.{i}.bias"].data_ptr()
    g.alpha_w, g.alpha_b = grads["alpha_linear.weight"].data_ptr(), grads["alpha_linear.bias"].data_ptr()
    g.rgb_w, g.rgb_b = grads["rgb_linear.weight"].data_ptr(), grads["rgb_linear.bias"].data_ptr()
    return g


# Arithmetic of the training forward: "bf16x6" (six bf16 piece products per fp32 product, weights and activations as
# three bf16 pieces: fp32-grade, 1.6x the fp32 pipe) or "f32" (fp32 MFMA).  The backward's pipes are a property of the
# library build.
TRAIN_PRECISION = os.environ.get("IDN_TRAIN_PRECISION", "bf16x6")


def _train_query(net, folded, rays, z):
    lib = _lib.load()
    ops._shape(z, "z", None, None)
    n, S = z.shape
    ops._shape(rays, "rays", n, ops.RAY_FLOATS)
    if TRAIN_PRECISION not in ("f32", "bf16x6"):
        raise _lib.IdealNerfError(f"IDN_TRAIN_PRECISION must be f32 or bf16x6 (got {TRAIN_PRECISION!r})")
    packed = net.packed_weights(TRAIN_PRECISION)
    code = _lib.IDN_PREC_F32 if TRAIN_PRECISION == "f32" else _lib.IDN_PREC_BF16X6
    with ops._Launch(packed, folded, rays, z) as L:
        raw = torch.empty((n, S, 4), dtype=torch.float32, device=z.device)
        acts = torch.empty(lib.idealnerf_train_acts_floats(n * S), dtype=torch.float32, device=z.device)
        check(lib.idealnerf_query_rays_train_fwd(ops._ptr(packed, "packed"), ops._ptr(folded, "folded"), code,
                                                 ops._ptr(rays, "rays"), ops._ptr(z, "z"), n, S, raw.data_ptr(),
                                                 acts.data_ptr(), L.stream))
    return raw, acts


def _pass_bwd(net, aud, expr, latent, acts, raw, z, rays, bc, g_rgb, g_fg, g_lw, g_acc, d_aud, d_latent):
    lib = _lib.load()
    n, S = z.shape
    sd = dict(net.named_parameters())
    grads = {k: torch.empty_like(sd[k]) for k in PARAM_KEYS}
    # columns the kernels never address (none today) would stay uninitialised: the fold
    # kernel writes all conditioning columns, the GEMM reductions all others.
    ps = net.kernel_params()
    gs = _grads_struct(grads)
    nbytes = lib.idealnerf_pass_bwd_workspace_bytes(n, S)
    for t, name, shape in ((raw, "raw", (n, S, 4)), (rays, "rays", (n, ops.RAY_FLOATS)), (bc, "bc_rgb", (n, 3)),
                           (g_rgb, "g_rgb_map", (n, 3)), (g_fg, "g_rgb_fg", (n, 3)), (g_lw, "g_last_weight", (n,)),
                           (g_acc, "g_acc", (n,))):
        ops._shape(t, name, *shape)
    ptr = ops._ptr
    with ops._Launch(aud, expr, latent, acts, raw, z, rays, bc, g_rgb, g_fg, g_lw, g_acc, d_aud, d_latent,
                     *grads.values()) as L:
        ws = ops._workspace(nbytes, z.device, L.stream)   # scratch of this (device, stream): two streams training on one GPU do not share it
        check(lib.idealnerf_pass_bwd(C.byref(ps), C.byref(gs), ptr(aud), ptr(expr), ptr(latent), ptr(acts),
                                     ptr(raw), ptr(z), ptr(rays), ptr(bc), n, S, ptr(g_rgb),
                                     ptr(g_fg), ptr(g_lw), ptr(g_acc), ptr(d_aud), ptr(d_latent), ws.data_ptr(),
                                     ws.numel(), L.stream))
    return grads


class RenderRaysFn(torch.autograd.Function):
    """outputs: rgb_map, disp_map, acc_map [, rgb0, disp0, acc0, z_std, last_weight]
    [, rgb_fg [, rgb_fg0, last_weight0]]  (middle group when N_importance > 0, last group for the
    torso variant).  disp / z_std carry no gradient."""

    N_FIXED = 13  # non-parameter arguments of forward

    @staticmethod
    def forward(ctx, coarse, fine, S, Ni, with_fg, rays, bc, expr, t_rand, u, aud, latent, lindisp, *params):
        dev = rays.device
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        aud_d, expr_d, lat_d = f32(aud), f32(expr), f32(latent)
        fc = coarse.folded_bias(aud_d, expr_d, lat_d)
        z_c = ops.coarse_depths(rays, linspace01(S, dev), t_rand, lindisp=lindisp)
        raw_c, acts_c = _train_query(coarse, fc, rays, z_c)
        if Ni > 0:   # coarse compositing + importance sampling + merge: one kernel, the weights stay on chip
            comp_c = smp = ops.march_fwd(raw_c, z_c, rays, bc, u, Ni, with_fg=with_fg)
        else:
            comp_c = ops.composite_fwd(raw_c, z_c, rays, bc, with_fg=with_fg, with_weights=False)
        # outputs the loss does not touch (acc_map, last_weight, ... in the reference's loop) come back as None, not as
        # zero tensors autograd would have to allocate and fill (a fill launch each): the C backward takes NULL for them
        ctx.set_materialize_grads(False)
        ctx.nets, ctx.Ni, ctx.with_fg = (coarse, fine), Ni, with_fg
        ctx.cond = (aud_d, expr_d, lat_d)
        ctx.needs = (aud is not None and aud.requires_grad, latent is not None and latent.requires_grad)
        if Ni == 0:
            ctx.saved = (rays, bc, raw_c, z_c, acts_c)
            outs = [comp_c["rgb_map"], comp_c["disp_map"], comp_c["acc_map"]]
            if with_fg:
                outs.append(comp_c["rgb_fg"])
            ctx.mark_non_differentiable(outs[1])
            return tuple(outs)
        ff = fine.folded_bias(aud_d, expr_d, lat_d)
        z_f = smp["z_fine"]
        raw_f, acts_f = _train_query(fine, ff, rays, z_f)
        comp_f = ops.composite_fwd(raw_f, z_f, rays, bc, with_fg=with_fg, with_weights=False)
        ctx.saved = (rays, bc, raw_c, z_c, acts_c, raw_f, z_f, acts_f)
        outs = [comp_f["rgb_map"], comp_f["disp_map"], comp_f["acc_map"], comp_c["rgb_map"], comp_c["disp_map"],
                comp_c["acc_map"], smp["z_std"], comp_f["last_weight"]]
        if with_fg:
            outs += [comp_f["rgb_fg"], comp_c["rgb_fg"], comp_c["last_weight"]]
        ctx.mark_non_differentiable(outs[1], outs[4], outs[6])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        coarse, fine = ctx.nets
        Ni, with_fg = ctx.Ni, ctx.with_fg
        aud, expr, lat = ctx.cond
        d_aud = torch.zeros_like(aud) if aud is not None else None
        d_lat = torch.zeros_like(lat) if lat is not None else None
        c = lambda t: None if t is None else t.contiguous()
        if Ni == 0:
            rays, bc, raw_c, z_c, acts_c = ctx.saved
            gc = _pass_bwd(coarse, aud, expr, lat, acts_c, raw_c, z_c, rays, bc, c(g[0]), c(g[3]) if with_fg else None,
                           None, c(g[2]), d_aud, d_lat)
            gf = {k: None for k in PARAM_KEYS}
        else:
            rays, bc, raw_c, z_c, acts_c, raw_f, z_f, acts_f = ctx.saved
            gf = _pass_bwd(fine, aud, expr, lat, acts_f, raw_f, z_f, rays, bc, c(g[0]), c(g[8]) if with_fg else None,
                           c(g[7]), c(g[2]), d_aud, d_lat)
            gc = _pass_bwd(coarse, aud, expr, lat, acts_c, raw_c, z_c, rays, bc, c(g[3]), c(g[9]) if with_fg else None,
                           c(g[10]) if with_fg else None, c(g[5]), d_aud, d_lat)
        ctx.saved = None
        need_aud, need_lat = ctx.needs
        param_grads = [gc[k] for k in PARAM_KEYS] + [gf[k] for k in PARAM_KEYS]
        fixed = [None] * RenderRaysFn.N_FIXED
        fixed[10] = d_aud if need_aud else None
        fixed[11] = d_lat if need_lat else None
        return (*fixed, *param_grads)


def render_rays_apply(network, coarse, fine, rays, bc_rgb, aud_para, latent_code, expr, perturb, pytest, with_fg=False,
                      lindisp=False):
    args = network.args
    rays = rays.detach().to(torch.float32).contiguous()
    bc = bc_rgb.detach().to(torch.float32).contiguous()
    n, dev = rays.shape[0], rays.device
    S, Ni = args.N_samples, args.N_importance
    t_rand, u = network.draw_randoms(n, S, Ni, perturb, pytest, dev)
    pc, pf = dict(coarse.named_parameters()), dict(fine.named_parameters())
    params = [pc[k] for k in PARAM_KEYS] + [pf[k] for k in PARAM_KEYS]
    outs = RenderRaysFn.apply(coarse, fine, S, Ni, with_fg, rays, bc, expr, t_rand, u, aud_para, latent_code, bool(lindisp),
                              *params)
    ret = {'rgb_map': outs[0], 'disp_map': outs[1], 'acc_map': outs[2]}
    if Ni > 0:
        ret.update(rgb0=outs[3], disp0=outs[4], acc0=outs[5], z_std=outs[6], last_weight=outs[7])
        if with_fg:
            ret.update(rgb_map_fg=outs[8], rgb_map_fg0=outs[9], last_weight0=outs[10])
    elif with_fg:
        ret['rgb_map_fg'] = outs[3]
    return ret


def facenerf_apply(module, x, aud, expr, latent_code):
    raise NotImplementedError(
        "FaceNeRF.forward with gradients on pre-embedded rows is not built: train through Network.render_rays "
        "(the reference's training path, audio_exp_nerf.py:534), or call under torch.no_grad() for inference")
