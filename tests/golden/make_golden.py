#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

What it does: registers empty stand-in modules for the
optional imports the hot path never touches (configargparse -> argparse,
face_alignment, imageio, natsort, cv2, tensorboard), makes ``Tensor.cuda`` a
no-op, supplies the flags through ``sys.argv`` before import (the reference
parses them at import time), then calls the reference's own functions on seeded
inputs and stores inputs + outputs as small ``.npz`` fixtures next to this file.

Weights are NOT stored: both this script and the tests rebuild them from
``oracle.xavier_facenerf_params(seed, dims)`` (numpy RandomState, Xavier-uniform,
bias 0.01 as audio_exp_nerf.py:442-448), optionally with the sigma head scaled
(``sigma_gain``) so the volume is not empty.

The torso variant (``rgb_map_fg``, TorsoNeRF/run_nerf.py:715-766, and the head + torso
composite, TorsoNeRF/train_torso.py:198-271) comes from the reference's own code as well:
``NeRFs.TorsoNeRF.run_nerf`` imports with the same stand-ins once ``F`` (which the module
uses without importing it) is put into its namespace, and ``NeRFs.TorsoNeRF.train_torso``
once ``get_embedder``'s default device is 'cpu' instead of 'cuda' (it only places the
frequency table) -- ``python make_golden.py headtorso``.  ``python make_golden.py frame512``
renders the first 4096 rays of the 512x512 bench frame (BASELINE configs[1]) with the
reference's ``Network.render_rays``.  ``python make_golden.py smoother`` / ``ds29`` run the reference's
``Network.forward`` itself on a 12 x 12 frame: behind ``nosmo_iters`` (the eight-frame audio window with its padding,
AudioNet + AudioAttNet) and in the ``dim_aud = 29`` configuration (``ds_aud_net``, FaceNeRFs with 29 audio columns).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"

NEAR, FAR = 0.5772005200386048, 1.1772005200386046


def install_shims():
    class _CfgParser(argparse.ArgumentParser):
        def add_argument(self, *a, **k):
            k.pop("is_config_file", None)
            return super().add_argument(*a, **k)

    m = types.ModuleType("configargparse")
    m.ArgumentParser = _CfgParser
    sys.modules["configargparse"] = m
    for name in ("face_alignment", "imageio", "cv2"):
        sys.modules[name] = types.ModuleType(name)
    ns = types.ModuleType("natsort")
    ns.natsorted = sorted
    sys.modules["natsort"] = ns
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # no-op
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, _):
            return lambda *a, **k: None

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    sys.modules["tensorboard"] = types.ModuleType("tensorboard")
    torch.Tensor.cuda = lambda self, *a, **k: self


def torso_raw2outputs():
    """NeRFs/TorsoNeRF/run_nerf.py::raw2outputs (:715-766), the reference's own function object.  The module
    imports with the stand-ins of install_shims(); it uses `F.relu` without importing `F`, which is supplied."""
    import importlib
    for p in (REF, os.path.join(REF, "NeRFs", "TorsoNeRF")):   # the module's siblings are imported by bare name upstream
        if p not in sys.path:
            sys.path.insert(0, p)
    rn = importlib.import_module("NeRFs.TorsoNeRF.run_nerf")
    torch.autograd.set_detect_anomaly(False)   # run_nerf_helpers switches it on at import (:7)
    rn.F = torch.nn.functional
    return rn.raw2outputs


def load_state(module, params):
    module.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)


def scale_sigma(params, gain, bias):
    p = {k: v.clone() for k, v in params.items()}
    p["alpha_linear.weight"] = p["alpha_linear.weight"] * gain
    p["alpha_linear.bias"] = torch.full_like(p["alpha_linear.bias"], bias)
    return p


def main():
    install_shims()
    sys.argv = [sys.argv[0], "--perturb", "0", "--dim_aud", "64", "--dim_expr", "76",
                "--N_samples", "64", "--N_importance", "128", "--near", str(NEAR), "--far", str(FAR),
                "--vis_path", "/tmp/idealnerf_golden_vis", "--chunk", "512"]
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    import oracle  # only for weight construction + synthetic inputs (no oracle output is stored)
    from models.face_nerf import FaceNeRF
    import NeRFs.HeadNeRF.helper as helper
    from NeRFs.HeadNeRF.train import baseline
    from NeRFs.HeadNeRF.train import audio_exp_nerf as aen

    rs = np.random.RandomState(1234)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    out = {}

    # ---- a2: positional encoding -------------------------------------------------
    x = f32(rs.uniform(-1.2, 1.2, size=(256, 3)))
    e10, d10 = helper.get_embedder(10, 0)
    e4, d4 = helper.get_embedder(4, 0)
    e3, d3 = helper.get_embedder(3, 0)
    np.savez(os.path.join(HERE, "pe.npz"), x=x.numpy(), pe10=e10(x).numpy(), pe4=e4(x).numpy(),
             pe3=e3(x).numpy())
    assert (d10, d4, d3) == (63, 27, 21)

    # ---- a5: FaceNeRF, C in {235, 169, 127} ---------------------------------------
    variants = {"c235": dict(dim_aud=64, dim_expr=76, dim_latent=32),
                "c169": dict(dim_aud=106, dim_expr=0, dim_latent=0),
                "c127": dict(dim_aud=64, dim_expr=0, dim_latent=0)}
    fx = {}
    for name, v in variants.items():
        dims = oracle.facenerf_dims(**v)
        params = oracle.xavier_facenerf_params(11, dims)
        net = FaceNeRF(D=8, W=256, input_ch=63, input_ch_views=27, dim_aud=v["dim_aud"],
                       dim_latent=v["dim_latent"], dim_expr=v["dim_expr"], skips=[4])
        assert {k: tuple(t.shape) for k, t in net.state_dict().items()} == oracle.facenerf_param_shapes(dims)
        load_state(net, params)
        pts = f32(rs.uniform(-1.0, 1.0, size=(512, 3)))
        dirs = f32(rs.standard_normal((512, 3)))
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        xin = torch.cat([e10(pts), e4(dirs)], -1)
        aud = f32(rs.standard_normal(v["dim_aud"]))
        expr = f32(rs.standard_normal(v["dim_expr"])) if v["dim_expr"] else None
        lat = f32(rs.standard_normal(v["dim_latent"])) if v["dim_latent"] else None
        with torch.no_grad():
            y = net(xin, aud, expr, lat)
        fx[name + "_x"] = xin.numpy()
        fx[name + "_aud"] = aud.numpy()
        if expr is not None:
            fx[name + "_expr"] = expr.numpy()
        if lat is not None:
            fx[name + "_latent"] = lat.numpy()
        fx[name + "_out"] = y.numpy()
    np.savez(os.path.join(HERE, "facenerf.npz"), **fx)

    # ---- a6: raw2outputs, S in {64, 192} -----------------------------------------
    fx = {}
    for S in (64, 192):
        n = 64
        raw = f32(rs.standard_normal((n, S, 4)))
        raw[..., 3] = raw[..., 3] * 40.0  # mix of empty and dense samples
        z = torch.sort(f32(rs.uniform(NEAR, FAR, size=(n, S))), dim=-1)[0]
        d = f32(rs.standard_normal((n, 3)))
        bc = f32(rs.uniform(0, 1, size=(n, 3)))
        with torch.no_grad():
            rgb_map, disp, acc, w, depth = baseline.raw2outputs(raw, z, d, bc)
            # the torso variant, by the reference's own function (TorsoNeRF/run_nerf.py:715-766): same five outputs + rgb_map_fg
            t_rgb_map, t_disp, t_acc, t_w, t_depth, rgb_fg = torso_raw2outputs()(raw, z, d, bc)
            assert all(torch.equal(a, b) for a, b in ((rgb_map, t_rgb_map), (disp, t_disp), (acc, t_acc), (w, t_w), (depth, t_depth)))
        fx.update({f"s{S}_raw": raw.numpy(), f"s{S}_z": z.numpy(), f"s{S}_d": d.numpy(), f"s{S}_bc": bc.numpy(),
                   f"s{S}_rgb_map": rgb_map.numpy(), f"s{S}_disp": disp.numpy(), f"s{S}_acc": acc.numpy(),
                   f"s{S}_weights": w.numpy(), f"s{S}_depth": depth.numpy(), f"s{S}_rgb_fg": rgb_fg.numpy()})
    np.savez(os.path.join(HERE, "raw2outputs.npz"), **fx)

    # ---- a7: sample_pdf, capturing cdf + inds at the searchsorted boundary ---------
    captured = {}
    real_ss = torch.searchsorted

    def spy(cdf, u, **kw):
        r = real_ss(cdf, u, **kw)
        captured["cdf"], captured["u"], captured["inds"] = cdf.clone(), u.clone(), r.clone()
        return r

    fx = {}
    n = 64
    z = torch.sort(f32(rs.uniform(NEAR, FAR, size=(n, 64))), dim=-1)[0]
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    w = f32(rs.uniform(0, 1, size=(n, 62))) ** 4
    w[:8] = 0.0          # empty rays: pdf uniform, denom guard
    w[8:16, 5:] = 0.0    # mass concentrated at the front
    torch.searchsorted = spy
    try:
        with torch.no_grad():
            zs_det = helper.sample_pdf(bins, w, 128, det=True)
        fx.update(bins=bins.numpy(), weights=w.numpy(), det_cdf=captured["cdf"].numpy(),
                  det_u=captured["u"].numpy(), det_inds=captured["inds"].numpy(), det_samples=zs_det.numpy())
        with torch.no_grad():
            zs_rnd = helper.sample_pdf(bins, w, 128, det=False, pytest=True)  # numpy seed 0 u
        fx.update(rnd_cdf=captured["cdf"].numpy(), rnd_u=captured["u"].numpy(),
                  rnd_inds=captured["inds"].numpy(), rnd_samples=zs_rnd.numpy())
    finally:
        torch.searchsorted = real_ss
    np.savez(os.path.join(HERE, "sample_pdf.npz"), **fx)

    # ---- a10: per-frame audio nets (models/audio_net.py); weights stored (small, torch-initialised)
    from models.audio_net import AudioNet, AudioAttNet, DeepSpeechAudNet
    torch.manual_seed(77)
    fx = {}
    auds = f32(rs.standard_normal((8, 16, 29)))
    nets = {"aud": AudioNet(64, 16), "att": AudioAttNet(), "ds": DeepSpeechAudNet()}
    for tag, m in nets.items():
        m.apply(aen.init_weights)
        for k, v in m.state_dict().items():
            fx[f"{tag}_sd_{k}"] = v.numpy()
    with torch.no_grad():
        out8 = nets["aud"](auds)          # [8, 64]: the smoothing window (audio_exp_nerf.py:256)
        out1 = nets["aud"](auds[3:4])     # [64]: the single-frame path (:259)
        att = nets["att"](out8)           # [64]  (:257)
        ds = nets["ds"](auds[3:4])        # [29]  (:262)
    fx.update(aud_in=auds.numpy(), aud_out8=out8.numpy(), aud_out1=out1.numpy(), att_out=att.numpy(), ds_out=ds.numpy())
    np.savez_compressed(os.path.join(HERE, "audio_nets.npz"), **fx)

    # ---- a3-a9: Network.render_rays (+ a1/a9 full-frame harness, a12 train harness) --
    dims = oracle.facenerf_dims()
    H = W = 32
    syn = oracle.synthetic_frame(H, W, seed=0, dims=dims)
    net = aen.Network(H, W, syn["focal"], NEAR, FAR, 512, None, 64, 128)
    pc = scale_sigma(oracle.xavier_facenerf_params(2, dims), 300.0, 0.3)
    pf = scale_sigma(oracle.xavier_facenerf_params(3, dims), 300.0, 0.3)
    load_state(net.face_nerf_coarse, pc)
    load_state(net.face_nerf_fine, pf)
    net.eval()

    ro, rd = helper.get_rays(H, W, syn["focal"], syn["c2w"])
    taps = {}
    real_r2o, real_sp = aen.raw2outputs, aen.sample_pdf

    def tap_r2o(raw, z_vals, rays_d, bc_rgb, *a, **k):
        r = real_r2o(raw, z_vals, rays_d, bc_rgb, *a, **k)
        tag = "coarse" if raw.shape[1] == 64 else "fine"
        taps.setdefault("raw_" + tag, []).append(raw.detach().clone())
        taps.setdefault("z_" + tag, []).append(z_vals.detach().clone())
        taps.setdefault("weights_" + tag, []).append(r[3].detach().clone())
        return r

    def tap_sp(bins, weights, N, det=False, pytest=False):
        torch.searchsorted = spy
        try:
            r = real_sp(bins, weights, N, det=det, pytest=pytest)
        finally:
            torch.searchsorted = real_ss
        taps.setdefault("cdf", []).append(captured["cdf"])
        taps.setdefault("u", []).append(captured["u"])
        taps.setdefault("inds", []).append(captured["inds"])
        taps.setdefault("z_samples", []).append(r.detach().clone())
        return r

    aen.raw2outputs, aen.sample_pdf = tap_r2o, tap_sp
    try:
        # (i) full frame through render_dynamic_face (rays from the pose; chunk=512 -> 2 chunks)
        with torch.no_grad():
            rgb, disp, acc, last_w, extras = net.render_dynamic_face(
                H, W, syn["focal"], expr=syn["expr"], poses=syn["c2w"], latent_code=syn["latent"],
                render_poses=syn["c2w"][:3, :4], chunk=512, near=NEAR, far=FAR, rays=None, bc_rgb=syn["bc"],
                aud_para=syn["aud"], ndc=False)
        viewdirs = rd / torch.norm(rd, dim=-1, keepdim=True)
        rays = torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), NEAR * torch.ones(H * W, 1),
                          FAR * torch.ones(H * W, 1), viewdirs.reshape(-1, 3)], -1).float()
        cat = lambda k: torch.cat(taps[k], 0).numpy()
        fx = dict(rays=rays.numpy(), rgb=rgb.numpy(), disp=disp.numpy(), acc=acc.numpy(), last_weight=last_w.numpy(),
                  rgb0=extras["rgb0"].numpy(), disp0=extras["disp0"].numpy(), acc0=extras["acc0"].numpy(),
                  z_std=extras["z_std"].numpy(),
                  tap_z_coarse=cat("z_coarse"), tap_weights_coarse=cat("weights_coarse"),
                  tap_cdf=cat("cdf"), tap_u=cat("u")[:1], tap_inds=cat("inds").astype(np.int16),
                  tap_z_samples=cat("z_samples"), tap_z_fine=cat("z_fine"),
                  # the bulky per-sample taps only for the first 128 rays
                  tap_raw_coarse=cat("raw_coarse")[:128], tap_raw_fine=cat("raw_fine")[:128],
                  tap_weights_fine=cat("weights_fine")[:128])
        np.savez_compressed(os.path.join(HERE, "frame32.npz"), **fx)

        # (ii) 64 rays with the reference's numpy-seeded jitter (perturb=1, pytest=True)
        taps.clear()
        sel = torch.from_numpy(rs.choice(H * W, size=64, replace=False))
        with torch.no_grad():
            ret = net.render_rays(rays[sel], syn["bc"].reshape(-1, 3)[sel], syn["aud"], syn["c2w"], syn["latent"],
                                  syn["expr"], perturb=1.0, pytest=True)
        np.random.seed(0)
        t_rand = np.random.rand(64, 64).astype(np.float32)  # what the reference drew (audio_exp_nerf.py:324-326)
        fx = dict(sel=sel.numpy(), t_rand=t_rand, u=cat("u"), inds=cat("inds"), cdf=cat("cdf"),
                  z_coarse=cat("z_coarse"), z_samples=cat("z_samples"), z_fine=cat("z_fine"),
                  weights_coarse=cat("weights_coarse"), raw_coarse=cat("raw_coarse"), raw_fine=cat("raw_fine"),
                  **{k: v.numpy() for k, v in ret.items()})
        np.savez_compressed(os.path.join(HERE, "rays64_jitter.npz"), **fx)

        # (iii) config 1: 256 rays, N_importance = 0 (render_rays directly; a9 KeyError path avoided)
        taps.clear()
        aen.args.N_importance = 0
        sel0 = torch.from_numpy(rs.choice(H * W, size=256, replace=False))
        with torch.no_grad():
            ret0 = net.render_rays(rays[sel0], syn["bc"].reshape(-1, 3)[sel0], syn["aud"], syn["c2w"], syn["latent"],
                                   syn["expr"])
        aen.args.N_importance = 128
        np.savez_compressed(os.path.join(HERE, "rays256_coarse_only.npz"), sel=sel0.numpy(),
                            **{k: v.numpy() for k, v in ret0.items()})

        # (iv) a12: one training step's loss + grads on 64 rays (perturb=0), autograd through the reference
        taps.clear()
        net.train()
        for p_ in net.parameters():
            p_.grad = None
        aud = syn["aud"].clone().requires_grad_(True)
        lat = syn["latent"].clone().requires_grad_(True)
        tgt = f32(rs.uniform(0, 1, size=(64, 3)))
        ret = net.render_rays(rays[sel], syn["bc"].reshape(-1, 3)[sel], aud, syn["c2w"], lat, syn["expr"])
        img_loss = helper.img2mse(ret["rgb_map"], tgt)
        loss = img_loss + helper.img2mse(ret["rgb0"], tgt) + 10 * (torch.norm(lat) * 0.0005)
        loss.backward()
        g = lambda m, k: dict(m.named_parameters())[k].grad.numpy()
        fx = dict(sel=sel.numpy(), target=tgt.numpy(), loss=loss.detach().numpy(), img_loss=img_loss.detach().numpy(),
                  rgb_map=ret["rgb_map"].detach().numpy(), rgb0=ret["rgb0"].detach().numpy(),
                  g_aud=aud.grad.numpy(), g_latent=lat.grad.numpy())
        for tag, m in (("c", net.face_nerf_coarse), ("f", net.face_nerf_fine)):
            for k in ("pts_linears.0.weight", "pts_linears.0.bias", "pts_linears.5.weight", "pts_linears.7.weight",
                      "views_linears.0.weight", "views_linears.2.bias", "alpha_linear.weight", "alpha_linear.bias",
                      "rgb_linear.weight", "rgb_linear.bias"):
                fx[f"g_{tag}_{k}"] = g(m, k)
        np.savez_compressed(os.path.join(HERE, "train_step.npz"), **fx)
    finally:
        aen.raw2outputs, aen.sample_pdf = real_r2o, real_sp

    # ---- L3: the region-weighted ray sampler (GetData.sample_rays, audio_exp_nerf.py:134-195) ------
    os.makedirs(aen.args.vis_path, exist_ok=True)
    open(os.path.join(aen.args.vis_path, "torso.jpg"), "a").close()   # skips the debug image write
    Hs = Ws = 96
    ds = object.__new__(aen.GetData)
    ds.H, ds.W, ds.focal, ds.cx, ds.cy = Hs, Ws, 256.0, 48.0, 48.0
    ds.args = types.SimpleNamespace(N_rand=96, sample_rate=0.95)
    aen.args.mouth_rays, aen.args.torso_rays = 16, 8
    pose = syn["c2w"].numpy().astype(np.float64)
    rect = np.array([10, 8, 70, 80], dtype=np.int32)
    lms = rs.uniform(8, 88, size=(68, 2))
    lms[48:] = rs.uniform(42, 54, size=(20, 2))
    parse = np.zeros((Hs, Ws, 3), dtype=np.uint8)
    parse[76:, 6:90] = (255, 0, 0)          # torso label
    parse[:6] = (0, 0, 255)
    target = f32(rs.uniform(0, 1, size=(Hs, Ws, 3)))
    bcimg = torch.from_numpy(rs.uniform(0, 1, size=(Hs, Ws, 3)))   # float64 like imread()/255.0
    np.random.seed(7)
    with torch.no_grad():
        brays, tgt_s, bc_s = ds.sample_rays(pose[:3, :4], rect, target, bcimg, lms, parse.copy())
    np.savez_compressed(os.path.join(HERE, "sample_rays.npz"), pose=pose[:3, :4], rect=rect, landmark=lms, parse=parse,
                        target=target.numpy(), bc=bcimg.numpy(), batch_rays=brays.numpy(), target_s=tgt_s.numpy(),
                        bc_s=bc_s.numpy(), focal=256.0, cx=48.0, cy=48.0, N_rand=96, mouth_rays=16, torso_rays=8,
                        sample_rate=0.95, seed=7)
    aen.args.mouth_rays, aen.args.torso_rays = 0, 0

    # ---- L4: config files through the reference's own flag parser -------------------------
    # (configargparse is absent; it turns `key = value` lines into `--key value` arguments, which is
    # what is fed to the reference's argparse-based parser here.)  Stored: the resolved values, or the
    # fact that the parser rejected the file (stale keys), for a handful of shipped configs.
    import glob
    import io
    import contextlib
    import json
    cfg_out = {}
    for path in sorted(glob.glob(os.path.join(REF, "NeRFs/HeadNeRF/configs/audio_expr_nerf/*/*.txt")) +
                       glob.glob(os.path.join(REF, "NeRFs/HeadNeRF/configs/audio_expr_nerf/*/*/*.txt"))):
        lines = [l.strip() for l in open(path) if l.strip() and l.strip()[0] not in "#;"]
        argv = []
        for l in lines:
            if "=" in l:
                k, v = [t.strip() for t in l.split("=", 1)]
                argv += ["--" + k, v]
            else:
                argv += ["--" + l]  # a bare key is a switch
        rel = os.path.relpath(path, REF)
        try:
            with contextlib.redirect_stderr(io.StringIO()):
                ns = helper.config_parser().parse_args(argv)
            cfg_out[rel] = {"lines": lines, "parsed": {k: v for k, v in vars(ns).items() if k != "config"}}
        except SystemExit:
            cfg_out[rel] = {"lines": lines, "parsed": None}
    with open(os.path.join(HERE, "configs_parsed.json"), "w") as f:
        json.dump(cfg_out, f, indent=0, sort_keys=True)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f:28s} {os.path.getsize(os.path.join(HERE, f)) / 1024:8.1f} KiB")


def agg_state(seed):
    """FaceNeRFAgg weights (models/face_nerf_agg.py:28-48 shapes, dim_agg=64, aud 64, expr 76, latent 32):
    Xavier-uniform from numpy RandomState(seed) in state_dict order, bias 0.01.  Rebuilt identically by
    the tests, so the fixture stores no weights."""
    import collections
    shapes = collections.OrderedDict()
    shapes["agg_linears.0"] = (64, 140)
    shapes["agg_linears.1"] = (64, 64)
    c_all = 63 + 64 + 32
    for i in range(8):
        shapes[f"pts_linears.{i}"] = (256, c_all if i == 0 else (256 + c_all if i == 5 else 256))
    shapes["views_linears.0"] = (128, 27 + 256 + 64)
    shapes["views_linears.1"] = (128, 128)
    shapes["views_linears.2"] = (128, 128)
    shapes["feature_linear"] = (256, 256)
    shapes["alpha_linear"] = (1, 256)
    shapes["rgb_linear"] = (3, 128)
    rs = np.random.RandomState(seed)
    sd = collections.OrderedDict()
    for k, (o, i) in shapes.items():
        bound = float(np.sqrt(6.0 / (o + i)))
        sd[k + ".weight"] = torch.from_numpy(rs.uniform(-bound, bound, size=(o, i)).astype(np.float32))
        sd[k + ".bias"] = torch.full((o,), 0.01, dtype=torch.float32)
    return sd


def golden_agg():
    """`python make_golden.py agg`: FaceNeRFAgg and the clip-level audio loop of
    NeRFs/TorsoNeRF/test_torso.py:478-498, written next to the other fixtures without touching them."""
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    torch.set_num_threads(8)
    from models.face_nerf_agg import FaceNeRFAgg
    from models.audio_net import AudioNet, AudioAttNet
    rs = np.random.RandomState(4321)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    net = FaceNeRFAgg(D=8, W=256, input_ch=63, input_ch_views=27, dim_agg=64, dim_aud=64, dim_expr=76, dim_latent=32,
                      skips=[4])
    sd = agg_state(21)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    x = f32(rs.uniform(-1.0, 1.0, size=(384, 90)))
    aud, expr, lat = f32(rs.standard_normal(64)), f32(rs.standard_normal(76)), f32(rs.standard_normal(32))
    with torch.no_grad():
        y = net(x, aud, expr, lat)
    np.savez(os.path.join(HERE, "facenerf_agg.npz"), x=x.numpy(), aud=aud.numpy(), expr=expr.numpy(), latent=lat.numpy(),
             out=y.numpy(), seed=21)

    # clip-level audio features: the literal loop of test_torso.py:478-498 (smo_size 8)
    torch.manual_seed(77)
    aud_net, att_net = AudioNet(64, 16), AudioAttNet()
    auds = f32(rs.standard_normal((21, 16, 29)))
    half, F = 4, auds.shape[0]
    outs = []
    with torch.no_grad():
        for i in range(F):
            left_i, right_i, pad_left, pad_right = i - half, i + half, 0, 0
            if left_i < 0:
                pad_left, left_i = -left_i, 0
            if right_i > F:
                pad_right, right_i = right_i - F, F
            win = auds[left_i:right_i]
            if pad_left > 0:
                win = torch.cat((torch.zeros_like(win)[:pad_left], win), dim=0)
            if pad_right > 0:
                win = torch.cat((win, torch.zeros_like(win)[:pad_right]), dim=0)
            outs.append(att_net(aud_net(win)))
    fx = {"auds": auds.numpy(), "out": torch.stack(outs, 0).numpy()}
    for k, v in aud_net.state_dict().items():
        fx["audnet." + k] = v.numpy()
    for k, v in att_net.state_dict().items():
        fx["attnet." + k] = v.numpy()
    np.savez(os.path.join(HERE, "audio_clip.npz"), **fx)
    print("wrote facenerf_agg.npz, audio_clip.npz")


def golden_torso():
    """`python make_golden.py torso`: the torso conditioning from the reference's own
    functions -- NeRFs/TorsoNeRF/run_nerf_helpers.py imports with numpy + torch alone.  pose_to_euler_trans
    (:26-47) on a batch of seeded poses, and the 106-d torso signal assembled per pose exactly as
    train_torso.py:238-240 does with the reference's get_embedder(3, 0) (:38; device='cpu' instead of its
    'cuda' default, which only places the frequency table)."""
    sys.path.insert(0, REF)
    import importlib
    H = importlib.import_module("NeRFs.TorsoNeRF.run_nerf_helpers")
    torch.autograd.set_detect_anomaly(False)   # the module switches it on at import (:7)
    embed_torso_aud_fn, ch = H.get_embedder(3, 0, device="cpu")
    assert ch == 21
    rs = np.random.RandomState(2024)
    B, dim_aud_body = 16, 64
    poses = np.zeros((B, 4, 4), dtype=np.float32)
    for b in range(B):
        a = rs.uniform(-0.6, 0.6, size=3)
        cx, cy, cz = np.cos(a)
        sx, sy, sz = np.sin(a)
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        poses[b, :3, :3] = (Rz @ Ry @ Rx).astype(np.float32)
        poses[b, :3, 3] = rs.uniform(-1.0, 1.0, size=3).astype(np.float32) * np.array([0.2, 0.2, 1.0], dtype=np.float32)
        poses[b, 3, 3] = 1.0
    aud = rs.standard_normal((B, 80)).astype(np.float32)      # wider than dim_aud_body: the slice is part of the formula
    tp, ta = torch.from_numpy(poses), torch.from_numpy(aud)
    with torch.no_grad():
        et_batch = H.pose_to_euler_trans(tp)
        signals = []
        for b in range(B):
            pose, aud_feature = tp[b], ta[b]
            et = H.pose_to_euler_trans(pose.unsqueeze(0))
            embed_et = torch.cat((embed_torso_aud_fn(et[:, :3]), embed_torso_aud_fn(et[:, 3:])), dim=1)
            signals.append(torch.cat((aud_feature[..., :dim_aud_body], torch.squeeze(embed_et)), dim=-1))
    np.savez(os.path.join(HERE, "torso_signal.npz"), poses=poses, aud=aud, euler_trans=et_batch.numpy(),
             signal=torch.stack(signals, 0).numpy(), dim_aud_body=dim_aud_body)
    print("wrote torso_signal.npz", et_batch.shape, torch.stack(signals, 0).shape)


def golden_flags():
    """`python make_golden.py flags`: the three switches the reference's Network leaves at their defaults
    (audio_exp_nerf.py:297-299) -- lindisp (:309-310), white_bkgd and raw_noise_std (baseline.py:353-373) --
    from the reference's own raw2outputs and render_rays on 64 rays of the 32x32 golden frame."""
    install_shims()
    sys.argv = [sys.argv[0], "--perturb", "0", "--dim_aud", "64", "--dim_expr", "76",
                "--N_samples", "64", "--N_importance", "128", "--near", str(NEAR), "--far", str(FAR),
                "--vis_path", "/tmp/idealnerf_golden_vis", "--chunk", "512"]
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import oracle
    import NeRFs.HeadNeRF.helper as helper
    from NeRFs.HeadNeRF.train import baseline
    from NeRFs.HeadNeRF.train import audio_exp_nerf as aen
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    fx = {}
    # raw2outputs on the S=64 inputs of raw2outputs.npz
    g = dict(np.load(os.path.join(HERE, "raw2outputs.npz")))
    raw, z, d, bc = (f32(g["s64_" + k]) for k in ("raw", "z", "d", "bc"))
    with torch.no_grad():
        for tag, kw in (("white", dict(white_bkgd=True)), ("noise", dict(raw_noise_std=0.7, pytest=True)),
                        ("both", dict(raw_noise_std=2.5, white_bkgd=True, pytest=True))):
            rgb_map, disp, acc, w, depth = baseline.raw2outputs(raw, z, d, bc, **kw)
            fx.update({f"r2o_{tag}_rgb_map": rgb_map.numpy(), f"r2o_{tag}_disp": disp.numpy(), f"r2o_{tag}_acc": acc.numpy(),
                       f"r2o_{tag}_weights": w.numpy(), f"r2o_{tag}_depth": depth.numpy()})
    # render_rays on 64 rays of the golden frame
    dims = oracle.facenerf_dims()
    H = W = 32
    syn = oracle.synthetic_frame(H, W, seed=0, dims=dims)
    net = aen.Network(H, W, syn["focal"], NEAR, FAR, 512, None, 64, 128)
    load_state(net.face_nerf_coarse, scale_sigma(oracle.xavier_facenerf_params(2, dims), 300.0, 0.3))
    load_state(net.face_nerf_fine, scale_sigma(oracle.xavier_facenerf_params(3, dims), 300.0, 0.3))
    net.eval()
    f = dict(np.load(os.path.join(HERE, "frame32.npz")))
    rays = f32(f["rays"])
    sel = torch.from_numpy(np.random.RandomState(99).choice(H * W, size=64, replace=False))
    fx["sel"] = sel.numpy()
    z_tap = {}
    real_r2o = aen.raw2outputs

    def tap(raw_, z_vals, *a, **k):
        z_tap.setdefault("coarse" if raw_.shape[1] == 64 else "fine", z_vals.detach().clone())
        return real_r2o(raw_, z_vals, *a, **k)

    aen.raw2outputs = tap
    try:
        with torch.no_grad():
            for tag, kw in (("lindisp", dict(lindisp=True)), ("white", dict(white_bkgd=True)),
                            ("noise", dict(raw_noise_std=0.5, pytest=True, perturb=0.0))):
                z_tap.clear()
                ret = net.render_rays(rays[sel], syn["bc"].reshape(-1, 3)[sel], syn["aud"], syn["c2w"], syn["latent"],
                                      syn["expr"], **kw)
                fx.update({f"rr_{tag}_{k}": v.numpy() for k, v in ret.items()})
                fx[f"rr_{tag}_z_coarse"] = z_tap["coarse"].numpy()
    finally:
        aen.raw2outputs = real_r2o
    np.savez_compressed(os.path.join(HERE, "flags.npz"), **fx)
    print("wrote flags.npz:", sorted(fx)[:6], "...", len(fx), "arrays")


def _spy_searchsorted(captured):
    real = torch.searchsorted

    def spy(cdf, u, **kw):
        r = real(cdf, u, **kw)
        captured["cdf"], captured["u"], captured["inds"] = cdf.clone(), u.clone(), r.clone()
        return r

    return real, spy


def golden_frame512():
    """`python make_golden.py frame512`: BASELINE configs[1] at full size, from the reference itself -- the first 4096
    rays (rows 0..7) of the 512x512 bench frame (oracle.synthetic_frame(512, 512, seed 0), networks seeds 2 / 3 with the
    density head x300, as bench.py and __graft_entry__.smoke build them) through the reference's Network.render_rays
    (audio_exp_nerf.py:297-371), perturb = 0.  Stored: every output for all 4096 rays, the importance indices of all
    4096 rays (captured at torch.searchsorted), and the merged fine depths of every 8th ray (512 rays) for the
    given-positions check of tests/parity_proof.py."""
    install_shims()
    sys.argv = [sys.argv[0], "--perturb", "0", "--dim_aud", "64", "--dim_expr", "76",
                "--N_samples", "64", "--N_importance", "128", "--near", str(NEAR), "--far", str(FAR),
                "--vis_path", "/tmp/idealnerf_golden_vis", "--chunk", "8192"]
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import oracle
    import NeRFs.HeadNeRF.helper as helper
    from NeRFs.HeadNeRF.train import audio_exp_nerf as aen
    dims = oracle.facenerf_dims()
    H = W = 512
    N = 4096
    syn = oracle.synthetic_frame(H, W, seed=0, dims=dims)
    net = aen.Network(H, W, syn["focal"], NEAR, FAR, 8192, None, 64, 128)
    load_state(net.face_nerf_coarse, scale_sigma(oracle.xavier_facenerf_params(2, dims), 300.0, 0.3))
    load_state(net.face_nerf_fine, scale_sigma(oracle.xavier_facenerf_params(3, dims), 300.0, 0.3))
    net.eval()
    ro, rd = helper.get_rays(H, W, syn["focal"], syn["c2w"])
    viewdirs = rd / torch.norm(rd, dim=-1, keepdim=True)
    rays = torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), NEAR * torch.ones(H * W, 1), FAR * torch.ones(H * W, 1),
                      viewdirs.reshape(-1, 3)], -1).float()[:N]
    bc = syn["bc"].reshape(-1, 3)[:N]
    captured, taps = {}, {}
    real_ss, spy = _spy_searchsorted(captured)
    real_r2o = aen.raw2outputs

    def tap_r2o(raw, z_vals, *a, **k):
        if raw.shape[1] == 192:
            taps["z_fine"] = z_vals.detach().clone()
        return real_r2o(raw, z_vals, *a, **k)

    torch.searchsorted, aen.raw2outputs = spy, tap_r2o
    try:
        with torch.no_grad():
            ret = net.render_rays(rays, bc, syn["aud"], syn["c2w"], syn["latent"], syn["expr"])
    finally:
        torch.searchsorted, aen.raw2outputs = real_ss, real_r2o
    inds = captured["inds"].numpy()
    assert inds.shape == (N, 128) and inds.min() >= 1 and inds.max() <= 63
    vis = float((ret["rgb_map"] - bc).abs().mean())
    fx = dict(n_rays=N, rays_first=rays[:4].numpy(), rays_last=rays[-4:].numpy(), inds=inds.astype(np.uint8),
              z_fine_every8=taps["z_fine"][::8].numpy(), **{k: v.numpy() for k, v in ret.items()})
    np.savez_compressed(os.path.join(HERE, "frame512_tile.npz"), **fx)
    print(f"wrote frame512_tile.npz: {N} rays, visibility {vis:.3f}, "
          f"{os.path.getsize(os.path.join(HERE, 'frame512_tile.npz')) / 1024:.0f} KiB")


def head_torso_inputs(n=512):
    """The seeded inputs of the head + torso scene (shared with tests/test_hip_parity.py::_torso_setup: same seeds, same
    draws, in the same order)."""
    import oracle
    rs = np.random.RandomState(5)
    syn = oracle.synthetic_frame(32, 32, seed=4)
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
    pose0 = torch.eye(4)
    pose0[:3, 3] = torch.tensor([0.02, -0.01, 0.9])
    ro, rd = oracle.camera_rays(32, 32, syn["focal"], pose[:3, :4])
    ro0, rd0 = oracle.camera_rays(32, 32, syn["focal"], pose0[:3, :4])
    sel = torch.from_numpy(rs.choice(1024, size=n, replace=False))
    pick = lambda a: a.reshape(-1, 3)[sel]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return syn, dict(batch_rays=torch.stack([pick(ro), pick(rd)], 0), batch_rays_torso=torch.stack([pick(ro0), pick(rd0)], 0),
                     bg=pick(syn["bc"]), auds=T(rs.standard_normal((4, 16, 29)).astype(np.float32)), pose=pose,
                     expr=T(rs.standard_normal(79).astype(np.float32)), latent=torch.ones(32),
                     target=T(rs.uniform(0, 1, (n, 3)).astype(np.float32)))


def golden_head_torso():
    """`python make_golden.py headtorso`: SURVEY 8 row a11 / BASELINE configs[4] from the reference's own
    NeRFs/TorsoNeRF/train_torso.py::Network (:198-271) -- forward([x, global_step, dataset_size]) in training mode (the
    ray-batch branch) under torch.no_grad(): head pair (C = 63 + 64 + 79 + 32) and torso pair (C = 63 + 106) rendered by
    its render_rays / run_nerf.raw2outputs / run_nerf_helpers.sample_pdf, composited as rgb * last_weight_torso +
    rgb_fg_torso (:269-270).  512 rays of the sharp test scene (density head x100 on the head pair, x4 on the torso
    pair).  Weights are rebuilt from seeds by the tests; the audio net's (torch-initialised) are stored."""
    install_shims()
    sys.argv = [sys.argv[0], "--perturb", "0", "--dim_aud", "64", "--dim_aud_body", "64", "--N_samples", "64",
                "--N_importance", "128", "--near", str(NEAR), "--far", str(FAR), "--chunk", "512",
                "--vis_path", "/tmp/idealnerf_golden_vis"]
    for p in (REPO, REF, os.path.join(REF, "NeRFs", "TorsoNeRF")):
        sys.path.insert(0, p)
    torch.manual_seed(4321)
    torch.set_num_threads(8)
    import functools
    import importlib
    import oracle
    helpers = importlib.import_module("NeRFs.TorsoNeRF.run_nerf_helpers")
    torch.autograd.set_detect_anomaly(False)
    real_get_embedder = helpers.get_embedder
    helpers.get_embedder = functools.wraps(real_get_embedder)(lambda multires, i=0, device="cpu": real_get_embedder(multires, i, device))
    raw2outputs_ref = torso_raw2outputs()      # imports NeRFs.TorsoNeRF.run_nerf and supplies its missing `F`
    tt = importlib.import_module("NeRFs.TorsoNeRF.train_torso")
    assert tt.raw2outputs is raw2outputs_ref and tt.args.N_importance == 128 and tt.args.perturb == 0.0

    syn, d = head_torso_inputs(512)
    n = d["bg"].shape[0]
    net = tt.Network(32, 32, syn["focal"], NEAR, FAR, 512, 64, 128)
    dh = oracle.facenerf_dims(dim_aud=64, dim_expr=79, dim_latent=32)
    dt = oracle.facenerf_dims(dim_aud=106, dim_expr=0, dim_latent=0)
    load_state(net.face_nerf_coarse, scale_sigma(oracle.xavier_facenerf_params(21, dh), 100.0, 0.2))
    load_state(net.face_nerf_fine, scale_sigma(oracle.xavier_facenerf_params(22, dh), 100.0, 0.2))
    load_state(net.torso_coarse_nerf, scale_sigma(oracle.xavier_facenerf_params(23, dt), 4.0, -0.2))
    load_state(net.torso_fine_nerf, scale_sigma(oracle.xavier_facenerf_params(24, dt), 4.0, -0.2))
    net.train()

    captured, order = {}, []
    real_ss, spy = _spy_searchsorted(captured)
    real_sp, real_rr = tt.sample_pdf, net.render_rays
    taps = {}

    def tap_sp(bins, weights, N, det=False, pytest=False):
        torch.searchsorted = spy
        try:
            r = real_sp(bins, weights, N, det=det, pytest=pytest)
        finally:
            torch.searchsorted = real_ss
        order.append((captured["inds"].clone(), r.detach().clone()))
        return r

    def tap_rr(rays, *a, **k):     # keep every render's own outputs (forward only returns the composites)
        r = real_rr(rays, *a, **k)
        taps.setdefault("renders", []).append({key: v.detach().clone() for key, v in r.items()})
        taps.setdefault("rays", []).append(rays.detach().clone())
        return r

    tt.sample_pdf, net.render_rays = tap_sp, tap_rr
    x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], torch.zeros(1, 32, 32, 3),
         d["pose"], d["expr"][None], d["latent"], torch.tensor([1]))
    try:
        with torch.no_grad():
            rgb_com, rgb_com0 = net([x, 0, 4])
    finally:
        tt.sample_pdf = real_sp
        del net.render_rays
    assert len(order) == 2 and len(taps["renders"]) == 2 and rgb_com.shape == (n, 3)
    head, torso = taps["renders"]
    # train_torso.py:269-270, once more from the parts the reference itself produced
    assert torch.equal(rgb_com, head["rgb_map"] * torso["last_weight"][..., None] + torso["rgb_map_fg"])
    assert torch.equal(rgb_com0, head["rgb0"] * torso["last_weight0"][..., None] + torso["rgb_map_fg0"])
    with torch.no_grad():
        aud_feature = net.aud_net(d["auds"][1:2])
    fx = dict(rgb_com=rgb_com.numpy(), rgb_com0=rgb_com0.numpy(), aud_feature=aud_feature.numpy(),
              rays_head=taps["rays"][0].numpy(), rays_torso=taps["rays"][1].numpy(),
              inds_head=order[0][0].numpy().astype(np.uint8), inds_torso=order[1][0].numpy().astype(np.uint8),
              z_samples_head=order[0][1].numpy(), z_samples_torso=order[1][1].numpy())
    for tag, r in (("head", head), ("torso", torso)):
        for k in ("rgb_map", "rgb_map_fg", "rgb0", "rgb_map_fg0", "last_weight", "last_weight0", "z_std", "disp_map", "acc_map"):
            fx[f"{tag}_{k}"] = r[k].numpy()
    for k, v in d.items():
        fx["in_" + k] = v.numpy()
    for k, v in net.aud_net.state_dict().items():
        fx["audnet." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "head_torso.npz"), **fx)
    e = (rgb_com - d["bg"]).abs().mean()
    print(f"wrote head_torso.npz: {n} rays, composite visibility {float(e):.3f}, "
          f"{os.path.getsize(os.path.join(HERE, 'head_torso.npz')) / 1024:.0f} KiB")


def golden_smoother():
    """`python make_golden.py smoother`: the reference's `Network.forward` (audio_exp_nerf.py:228-279) BEHIND `nosmo_iters` --
    the window of `smo_size` = 8 DeepSpeech frames around `index`, zero-padded at the clip's ends (:246-262), AudioNet on the
    window and AudioAttNet on its output -- in eval mode on a 12 x 12 frame of a 10-frame clip, for a frame at the clip's
    start (three padded slots on the left), one in the middle and one at its end (two on the right).  Stored: the inputs,
    the two audio nets' weights (they are drawn from torch's generator upstream), and per frame the audio feature the
    reference hands to its renderer (captured at `render_dynamic_face`) and the rendered frame."""
    install_shims()
    sys.argv = [sys.argv[0], "--perturb", "0", "--dim_aud", "64", "--dim_expr", "76",
                "--N_samples", "64", "--N_importance", "128", "--near", str(NEAR), "--far", str(FAR),
                "--vis_path", "/tmp/idealnerf_golden_vis", "--chunk", "512"]
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import oracle
    from NeRFs.HeadNeRF.train import audio_exp_nerf as aen
    dims = oracle.facenerf_dims()
    H = W = 12
    F_ = 10
    syn = oracle.synthetic_frame(H, W, seed=9, dims=dims)
    net = aen.Network(H, W, syn["focal"], NEAR, FAR, 512, None, 64, 128)
    net.apply(aen.init_weights)
    load_state(net.face_nerf_coarse, scale_sigma(oracle.xavier_facenerf_params(2, dims), 300.0, 0.3))
    load_state(net.face_nerf_fine, scale_sigma(oracle.xavier_facenerf_params(3, dims), 300.0, 0.3))
    net.eval()
    rs = np.random.RandomState(77)
    auds = torch.from_numpy(rs.standard_normal((F_, 16, 29)).astype(np.float32))
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
    fx = dict(auds=auds.numpy(), pose=pose.numpy(), bg=syn["bc"].numpy(), expr=syn["expr"].numpy(), latent=syn["latent"].numpy(),
              focal=np.float64(syn["focal"]), nosmo_iters=np.int64(aen.args.nosmo_iters), frames=np.array([1, 5, 8]))
    for k, v in net.aud_net.state_dict().items():
        fx["audnet." + k] = v.numpy()
    for k, v in net.aud_att_net.state_dict().items():
        fx["attnet." + k] = v.numpy()
    real_rdf = net.render_dynamic_face
    for idx in (1, 5, 8):
        seen = {}

        def spy(*a, **k):
            seen["aud"] = k["aud_para"].detach().clone()
            return real_rdf(*a, **k)

        net.render_dynamic_face = spy
        data = (torch.zeros(1, 2, 1, 3), torch.zeros(1, 3), syn["bc"][None], auds[None], torch.zeros(1, H, W, 3), pose[None],
                syn["expr"][None], syn["latent"], torch.tensor([idx]))
        with torch.no_grad():
            rgb, disp, acc, last_w, extras = net([data, int(aen.args.nosmo_iters), F_])
        fx[f"aud_feature_{idx}"] = seen["aud"].numpy()
        fx[f"rgb_{idx}"] = rgb.numpy()
        fx[f"rgb0_{idx}"] = extras["rgb0"].numpy()
    net.render_dynamic_face = real_rdf
    assert fx["aud_feature_1"].shape == (64,) and not np.allclose(fx["aud_feature_1"], fx["aud_feature_5"])
    vis = float(np.abs(fx["rgb_5"] - syn["bc"].numpy()).mean())
    np.savez_compressed(os.path.join(HERE, "smoother.npz"), **fx)
    print(f"wrote smoother.npz: frames 1, 5, 8 of a {F_}-frame clip, {H}x{W}, visibility {vis:.3f}, "
          f"{os.path.getsize(os.path.join(HERE, 'smoother.npz')) / 1024:.0f} KiB")


def golden_ds29():
    """`python make_golden.py ds29`: the reference's `Network.forward` in its `dim_aud = 29` configuration (the DeepSpeech-logit
    ablation: `*_adnerf_baseline`-style configs; audio_exp_nerf.py:265-269 takes `ds_aud_net`, a Linear(16, 1) squeeze over
    the window, instead of AudioNet, and the FaceNeRFs are built with 29 audio columns: C = 63 + 29 + 76 + 32 = 200) in eval
    mode on a 12 x 12 frame.  Stored: inputs, `ds_aud_net`'s weights, the audio feature handed to the renderer, the frame."""
    install_shims()
    sys.argv = [sys.argv[0], "--perturb", "0", "--dim_aud", "29", "--dim_expr", "76",
                "--N_samples", "64", "--N_importance", "128", "--near", str(NEAR), "--far", str(FAR),
                "--vis_path", "/tmp/idealnerf_golden_vis", "--chunk", "512"]
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import oracle
    from NeRFs.HeadNeRF.train import audio_exp_nerf as aen
    dims = oracle.facenerf_dims(dim_aud=29)
    H = W = 12
    syn = oracle.synthetic_frame(H, W, seed=9, dims=dims)
    net = aen.Network(H, W, syn["focal"], NEAR, FAR, 512, None, 64, 128)
    net.apply(aen.init_weights)
    load_state(net.face_nerf_coarse, scale_sigma(oracle.xavier_facenerf_params(2, dims), 300.0, 0.3))
    load_state(net.face_nerf_fine, scale_sigma(oracle.xavier_facenerf_params(3, dims), 300.0, 0.3))
    net.eval()
    rs = np.random.RandomState(78)
    auds = torch.from_numpy(rs.standard_normal((6, 16, 29)).astype(np.float32))
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
    fx = dict(auds=auds.numpy(), pose=pose.numpy(), bg=syn["bc"].numpy(), expr=syn["expr"].numpy(), latent=syn["latent"].numpy(),
              focal=np.float64(syn["focal"]), index=np.int64(4))
    for k, v in net.ds_aud_net.state_dict().items():
        fx["dsnet." + k] = v.numpy()
    real_rdf, seen = net.render_dynamic_face, {}

    def spy(*a, **k):
        seen["aud"] = k["aud_para"].detach().clone()
        return real_rdf(*a, **k)

    net.render_dynamic_face = spy
    data = (torch.zeros(1, 2, 1, 3), torch.zeros(1, 3), syn["bc"][None], auds[None], torch.zeros(1, H, W, 3), pose[None],
            syn["expr"][None], syn["latent"], torch.tensor([4]))
    with torch.no_grad():
        rgb, disp, acc, last_w, extras = net([data, 0, 6])
    assert seen["aud"].shape == (29,) and tuple(net.face_nerf_coarse.pts_linears[0].weight.shape) == (256, 200)
    fx.update(aud_feature=seen["aud"].numpy(), rgb=rgb.numpy(), rgb0=extras["rgb0"].numpy(), last_weight=last_w.numpy())
    vis = float(np.abs(fx["rgb"] - syn["bc"].numpy()).mean())
    np.savez_compressed(os.path.join(HERE, "ds29.npz"), **fx)
    print(f"wrote ds29.npz: {H}x{W}, visibility {vis:.3f}, {os.path.getsize(os.path.join(HERE, 'ds29.npz')) / 1024:.0f} KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "flags":
        golden_flags()
    elif len(sys.argv) > 1 and sys.argv[1] == "frame512":
        golden_frame512()
    elif len(sys.argv) > 1 and sys.argv[1] == "headtorso":
        golden_head_torso()
    elif len(sys.argv) > 1 and sys.argv[1] == "agg":
        golden_agg()
    elif len(sys.argv) > 1 and sys.argv[1] == "torso":
        golden_torso()
    elif len(sys.argv) > 1 and sys.argv[1] == "smoother":
        golden_smoother()
    elif len(sys.argv) > 1 and sys.argv[1] == "ds29":
        golden_ds29()
    else:
        main()
