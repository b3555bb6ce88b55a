"""Pin the CPU oracle against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import oracle

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
NEAR, FAR = 0.5772005200386048, 1.1772005200386046


def scale_sigma(p, gain=300.0, bias=0.3):
    p = {k: v.clone() for k, v in p.items()}
    p["alpha_linear.weight"] = p["alpha_linear.weight"] * gain
    p["alpha_linear.bias"] = torch.full_like(p["alpha_linear.bias"], bias)
    return p


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_positional_encoding(golden):
    g = golden("pe")
    x = T(g["x"])
    for L, key in ((10, "pe10"), (4, "pe4"), (3, "pe3")):
        out = oracle.positional_encoding(x, L).numpy()
        assert out.shape == g[key].shape
        np.testing.assert_array_equal(out, g[key])


@pytest.mark.parametrize("name,v", [("c235", dict(dim_aud=64, dim_expr=76, dim_latent=32)),
                                    ("c169", dict(dim_aud=106, dim_expr=0, dim_latent=0)),
                                    ("c127", dict(dim_aud=64, dim_expr=0, dim_latent=0))])
def test_facenerf(golden, name, v):
    g = golden("facenerf")
    dims = oracle.facenerf_dims(**v)
    p = oracle.xavier_facenerf_params(11, dims)
    opt = lambda k: T(g[k]) if k in g else None
    y = oracle.facenerf_forward(p, T(g[name + "_x"]), T(g[name + "_aud"]), opt(name + "_expr"),
                                opt(name + "_latent"), dims)
    assert rel_err(y.numpy(), g[name + "_out"]) < 2e-6


@pytest.mark.parametrize("S", [64, 192])
def test_composite(golden, S):
    g = golden("raw2outputs")
    k = lambda n: T(g[f"s{S}_{n}"])
    rgb, disp, acc, w, depth, fg = oracle.composite(k("raw"), k("z"), k("d"), k("bc"), with_fg=True)
    np.testing.assert_array_equal(w.numpy(), g[f"s{S}_weights"])
    np.testing.assert_array_equal(rgb.numpy(), g[f"s{S}_rgb_map"])
    np.testing.assert_array_equal(disp.numpy(), g[f"s{S}_disp"])
    np.testing.assert_array_equal(acc.numpy(), g[f"s{S}_acc"])
    np.testing.assert_array_equal(depth.numpy(), g[f"s{S}_depth"])
    np.testing.assert_array_equal(fg.numpy(), g[f"s{S}_rgb_fg"])


@pytest.mark.parametrize("mode", ["det", "rnd"])
def test_sample_importance(golden, mode):
    g = golden("sample_pdf")
    u = None if mode == "det" else T(g["rnd_u"])
    z, inds, cdf, u_used = oracle.sample_importance(T(g["bins"]), T(g["weights"]), 128, det=(mode == "det"), u=u)
    np.testing.assert_array_equal(u_used.numpy(), g[mode + "_u"])
    np.testing.assert_array_equal(cdf.numpy(), g[mode + "_cdf"])
    np.testing.assert_array_equal(inds.numpy(), g[mode + "_inds"])       # the bit-exact target
    np.testing.assert_array_equal(z.numpy(), g[mode + "_samples"])
    # the boundary on its own: golden (cdf, u) in -> golden inds out
    z2, inds2 = oracle.invert_cdf(T(g[mode + "_cdf"]), T(g["bins"]), T(g[mode + "_u"]))
    np.testing.assert_array_equal(inds2.numpy(), g[mode + "_inds"])
    np.testing.assert_array_equal(z2.numpy(), g[mode + "_samples"])


def _nets():
    dims = oracle.facenerf_dims()
    return dims, scale_sigma(oracle.xavier_facenerf_params(2, dims)), scale_sigma(oracle.xavier_facenerf_params(3, dims))


def test_frame32(golden):
    g = golden("frame32")
    dims, pc, pf = _nets()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    ro, rd = oracle.camera_rays(32, 32, syn["focal"], syn["c2w"])
    rays = oracle.ray_records(ro, rd, NEAR, FAR)
    np.testing.assert_array_equal(rays.numpy(), g["rays"])
    with torch.no_grad():
        out = oracle.render_frame(32, 32, syn["focal"], syn["c2w"], NEAR, FAR, syn["bc"], pc, pf, syn["aud"],
                                  syn["expr"], syn["latent"], chunk=512, dims=dims, taps=True)
    # same machine, same op sequence => the oracle reproduces the reference's index decisions exactly
    np.testing.assert_array_equal(out["tap_inds"].reshape(1024, 128).numpy(), g["tap_inds"].astype(np.int64))
    np.testing.assert_array_equal(out["tap_z_coarse"].reshape(1024, 64).numpy(), g["tap_z_coarse"])
    for k, gk, tol in (("rgb_map", "rgb", 1e-6), ("disp_map", "disp", 1e-6), ("acc_map", "acc", 1e-6),
                       ("last_weight", "last_weight", 1e-6), ("rgb0", "rgb0", 1e-6), ("z_std", "z_std", 1e-6),
                       ("tap_weights_coarse", "tap_weights_coarse", 1e-6), ("tap_cdf", "tap_cdf", 1e-6),
                       ("tap_z_fine", "tap_z_fine", 1e-6)):
        assert rel_err(out[k].reshape(g[gk].shape).numpy(), g[gk]) < tol, k
    # the frame is not empty: the volume contributes visibly more than the background alone
    assert np.abs(g["rgb"].reshape(-1, 3) - syn["bc"].reshape(-1, 3).numpy()).mean() > 0.05


def test_rays64_jitter(golden):
    g = golden("rays64_jitter")
    f = golden("frame32")
    dims, pc, pf = _nets()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    sel = T(g["sel"])
    rays, bc = T(f["rays"])[sel], syn["bc"].reshape(-1, 3)[sel]
    with torch.no_grad():
        out = oracle.render_rays(rays, bc, pc, pf, syn["aud"], syn["expr"], syn["latent"], dims=dims,
                                 t_rand=T(g["t_rand"]), u=T(g["u"]), taps=True)
    np.testing.assert_array_equal(out["tap_z_coarse"].numpy(), g["z_coarse"])
    np.testing.assert_array_equal(out["tap_inds"].numpy(), g["inds"])
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "z_std", "last_weight"):
        assert rel_err(out[k].numpy(), g[k]) < 1e-6, k
    assert rel_err(out["tap_z_fine"].numpy(), g["z_fine"]) < 1e-6


def test_rays256_coarse_only(golden):
    g = golden("rays256_coarse_only")
    f = golden("frame32")
    dims, pc, _ = _nets()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    sel = T(g["sel"])
    with torch.no_grad():
        out = oracle.render_rays(T(f["rays"])[sel], syn["bc"].reshape(-1, 3)[sel], pc, None, syn["aud"], syn["expr"],
                                 syn["latent"], n_importance=0, dims=dims)
    assert set(out) == {"rgb_map", "disp_map", "acc_map"}
    for k in out:
        assert rel_err(out[k].numpy(), g[k]) < 1e-6, k


def test_train_step(golden):
    g = golden("train_step")
    f = golden("frame32")
    dims, pc, pf = _nets()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    sel = T(g["sel"])
    for p in (pc, pf):
        for v in p.values():
            v.requires_grad_(True)
    aud = syn["aud"].clone().requires_grad_(True)
    lat = syn["latent"].clone().requires_grad_(True)
    out = oracle.render_rays(T(f["rays"])[sel], syn["bc"].reshape(-1, 3)[sel], pc, pf, aud, syn["expr"], lat, dims=dims)
    loss, img_loss = oracle.train_loss(out, T(g["target"]), lat)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    assert rel_err(aud.grad.numpy(), g["g_aud"]) < 1e-4
    assert rel_err(lat.grad.numpy(), g["g_latent"]) < 1e-4
    for tag, p in (("c", pc), ("f", pf)):
        for k in ("pts_linears.0.weight", "pts_linears.5.weight", "views_linears.0.weight", "alpha_linear.weight",
                  "rgb_linear.bias"):
            assert rel_err(p[k].grad.numpy(), g[f"g_{tag}_{k}"]) < 1e-4, (tag, k)
    assert pc["feature_linear.weight"].grad is None   # constructed, never used (face_nerf.py:34 vs :66)


def test_torso_signal_shape():
    syn = oracle.synthetic_frame(8, 8)
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
    s = oracle.torso_signal(syn["aud"], pose)
    assert s.shape == (64 + 42,)
    e = oracle.pose_to_euler_trans(pose[None])
    assert torch.allclose(e[0, 3:], syn["c2w"][:, 3])


def test_torso_conditioning_golden(golden):
    """a11 pinned: the oracle's pose_to_euler_trans / torso_signal against the reference's own functions
    (NeRFs/TorsoNeRF/run_nerf_helpers.py:26-47, get_embedder(3, 0) :102-120, assembly train_torso.py:238-240),
    run in the build container by tests/golden/make_golden.py torso.  Same machine, same libm: bit-exact."""
    g = golden("torso_signal")
    poses, aud = torch.from_numpy(g["poses"]), torch.from_numpy(g["aud"])
    np.testing.assert_array_equal(oracle.pose_to_euler_trans(poses).numpy(), g["euler_trans"])
    for b in range(poses.shape[0]):
        sig = oracle.torso_signal(aud[b], poses[b], dim_aud_body=int(g["dim_aud_body"]))
        assert sig.shape == (106,)
        np.testing.assert_array_equal(sig.numpy(), g["signal"][b])
    # the product's host-side restatement of the same three lines, on CPU tensors (plain torch math)
    from idealnerf_amd.train_torso import pose_to_euler_trans
    np.testing.assert_array_equal(pose_to_euler_trans(poses).numpy(), g["euler_trans"])


def test_render_switches_golden(golden):
    """lindisp / white_bkgd / raw_noise_std (the switches the reference's Network leaves at their defaults,
    audio_exp_nerf.py:297-299,309-310; baseline.py:353-373) in the oracle against the reference's own outputs."""
    g, r = golden("flags"), golden("raw2outputs")
    f32 = lambda k: torch.from_numpy(r["s64_" + k])
    raw, z, d, bc = f32("raw"), f32("z"), f32("d"), f32("bc")
    np.random.seed(0)
    u07 = torch.Tensor(np.random.rand(64, 64) * 0.7)
    np.random.seed(0)
    u25 = torch.Tensor(np.random.rand(64, 64) * 2.5)
    for tag, kw in (("white", dict(white_bkgd=True)), ("noise", dict(sigma_noise=u07)), ("both", dict(sigma_noise=u25, white_bkgd=True))):
        out = oracle.composite(raw, z, d, bc, **kw)
        for got, name in zip(out, ("rgb_map", "disp", "acc", "weights", "depth")):
            np.testing.assert_array_equal(got.numpy(), g[f"r2o_{tag}_{name}"], err_msg=f"{tag} {name}")
    near = torch.full((4, 1), 0.5772005200386048)
    far = torch.full((4, 1), 1.1772005200386046)
    np.testing.assert_array_equal(oracle.coarse_depths(near, far, 64, lindisp=True).numpy(), g["rr_lindisp_z_coarse"][:4])


def test_head_torso_golden(golden):
    """SURVEY 8 row a11 against the reference's own TorsoNeRF code (train_torso.py::Network.forward, run_nerf.raw2outputs
    with its rgb_map_fg, run_nerf_helpers.sample_pdf): the oracle's head render, torso render and composite."""
    g = golden("head_torso")
    dh = oracle.facenerf_dims(dim_aud=64, dim_expr=79, dim_latent=32)
    dt = oracle.facenerf_dims(dim_aud=106, dim_expr=0, dim_latent=0)
    P = dict(hc=scale_sigma(oracle.xavier_facenerf_params(21, dh), 100.0, 0.2), hf=scale_sigma(oracle.xavier_facenerf_params(22, dh), 100.0, 0.2),
             tc=scale_sigma(oracle.xavier_facenerf_params(23, dt), 4.0, -0.2), tf=scale_sigma(oracle.xavier_facenerf_params(24, dt), 4.0, -0.2))
    aud_feature = T(g["aud_feature"])
    with torch.no_grad():
        aud_torso = oracle.torso_signal(aud_feature, T(g["in_pose"]))
        rec = lambda r: oracle.ray_records(r[0], r[1], NEAR, FAR)
        np.testing.assert_array_equal(rec(T(g["in_batch_rays"])).numpy(), g["rays_head"])     # the reference's own ray records
        np.testing.assert_array_equal(rec(T(g["in_batch_rays_torso"])).numpy(), g["rays_torso"])
        head = oracle.render_rays(T(g["rays_head"]), T(g["in_bg"]), P["hc"], P["hf"], aud_feature, T(g["in_expr"]), T(g["in_latent"]),
                                  dims=dh, with_fg=True, taps=True)
        torso = oracle.render_rays(T(g["rays_torso"]), T(g["in_bg"]), P["tc"], P["tf"], aud_torso, None, None, dims=dt, with_fg=True, taps=True)
        rgb_com, rgb_com0 = oracle.head_torso_composite(head, torso)
    for tag, r in (("head", head), ("torso", torso)):
        flips = float((r["tap_inds"].numpy() != g[f"inds_{tag}"].astype(np.int64)).mean())
        assert flips < 2e-4, (tag, flips)     # same code on the same host class: 0 here; a BLAS difference may flip a few
        for k in ("rgb0", "rgb_map_fg0", "last_weight0"):     # nothing is sampled before the coarse pass
            assert rel_err(r[k].numpy(), g[f"{tag}_{k}"]) < 2e-6, (tag, k)
        same = (r["tap_inds"].numpy() == g[f"inds_{tag}"].astype(np.int64)).all(1)
        for k in ("rgb_map", "rgb_map_fg", "last_weight"):
            assert rel_err(r[k].numpy()[same], g[f"{tag}_{k}"][same]) < 5e-6, (tag, k)
    assert rel_err(rgb_com0.numpy(), g["rgb_com0"]) < 2e-6
    assert rel_err(rgb_com.numpy(), g["rgb_com"]) < 1e-4      # north_star's budget, end to end on the sharp scene


def test_frame512_tile_golden(golden):
    """BASELINE configs[1] at full size: the oracle on the first 4096 rays of the 512 x 512 bench frame against the
    reference's Network.render_rays (a sixth of the rays here to keep the CPU suite short; all of them in -m gpu)."""
    g = golden("frame512_tile")
    dims, pc, pf = _nets()
    syn = oracle.synthetic_frame(512, 512, seed=0, dims=dims)
    ro, rd = oracle.camera_rays(512, 512, syn["focal"], syn["c2w"])
    rays = oracle.ray_records(ro, rd, NEAR, FAR)[:int(g["n_rays"])]
    np.testing.assert_array_equal(rays[:4].numpy(), g["rays_first"])
    np.testing.assert_array_equal(rays[-4:].numpy(), g["rays_last"])
    idx = torch.arange(0, rays.shape[0], 6)
    with torch.no_grad():
        out = oracle.render_rays(rays[idx], syn["bc"].reshape(-1, 3)[idx], pc, pf, syn["aud"], syn["expr"], syn["latent"], dims=dims, taps=True)
    flips = float((out["tap_inds"].numpy() != g["inds"][idx.numpy()].astype(np.int64)).mean())
    assert flips < 2e-4, flips
    for k in ("rgb_map", "rgb0", "disp_map", "acc_map", "disp0", "acc0"):
        assert rel_err(out[k].numpy(), g[k][idx.numpy()]) < 1e-5, k
    on8 = (idx % 8 == 0)
    assert rel_err(out["tap_z_fine"][on8].numpy(), g["z_fine_every8"][(idx[on8] // 8).numpy()]) < 1e-6


@pytest.mark.parametrize("name,v", [("c235", dict(dim_aud=64, dim_expr=76, dim_latent=32)),
                                    ("c169", dict(dim_aud=106, dim_expr=0, dim_latent=0)),
                                    ("c127", dict(dim_aud=64, dim_expr=0, dim_latent=0))])
def test_bf16_rounding_model_is_the_reference_network_with_rounded_operands(golden, name, v):
    """`oracle.facenerf_forward_bf16_emulated` (the model the plain-bf16 kernel is held to in -m gpu) restated without its
    bias folding: the reference's own layer sequence (face_nerf.py:40-80, i.e. `facenerf_forward`) in float64 with the
    per-point columns of every weight and the per-point part of every layer input rounded to bf16, the per-frame
    conditioning columns left alone.  Both must agree to the fp32 rounding of the folded bias -- for all three
    conditioning layouts -- and sit where bf16 sits against the reference's fp32 output (6.5e-3, not 1e-5 and not 1e-1)."""
    g = golden("facenerf")
    dims = oracle.facenerf_dims(**v)
    p = oracle.xavier_facenerf_params(11, dims)
    opt = lambda k: T(g[f"{name}_{k}"]) if f"{name}_{k}" in g else None
    x, aud, expr, lat = T(g[name + "_x"]), opt("aud"), opt("expr"), opt("latent")
    bf = lambda t: t.to(torch.bfloat16).to(torch.float64)
    ci, cv, W = dims["input_ch"], dims["input_ch_views"], dims["W"]
    cond = torch.cat([t for t in (aud, None if expr is None else expr * 1 / 3, lat) if t is not None]).double()
    nc, c_all = cond.numel(), ci + cond.numel()
    q = {k: t.double() for k, t in p.items()}
    for k, cols in (("pts_linears.0", [(0, ci)]), ("pts_linears.5", [(0, ci), (c_all, c_all + W)]), ("views_linears.0", [(0, W + cv)])):
        for c0, c1 in cols:
            q[k + ".weight"][:, c0:c1] = bf(p[k + ".weight"][:, c0:c1])
    for k in [f"pts_linears.{i}" for i in (1, 2, 3, 4, 6, 7)] + ["views_linears.1", "views_linears.2", "alpha_linear", "rgb_linear"]:
        q[k + ".weight"] = bf(p[k + ".weight"])
    lin = lambda k, h: h @ q[k + ".weight"].t() + q[k + ".bias"]
    n = x.shape[0]
    initial = torch.cat([bf(x[:, :ci]), cond[None].expand(n, nc)], -1)
    h = initial
    for i in range(8):
        h = bf(torch.relu(lin(f"pts_linears.{i}", h)).float())
        if i == 4:
            h = torch.cat([initial, h], -1)
    sigma = lin("alpha_linear", h)
    parts = [h, bf(x[:, ci:])] + ([] if expr is None else [(expr * 1 / 3).double()[None].expand(n, -1)])
    h = torch.cat(parts, -1)
    for i in range(3):
        h = bf(torch.relu(lin(f"views_linears.{i}", h)).float())
    direct = torch.cat([lin("rgb_linear", h), sigma], -1)
    with torch.no_grad():
        emu = oracle.facenerf_forward_bf16_emulated(p, x, aud, expr, lat, dims).double()
    scale = float(np.abs(g[name + "_out"]).max())
    # the two differ where an fp32-rounded pre-activation (emu rounds each layer's output to fp32 before its bf16
    # rounding, as the kernel's accumulator does) falls on the other side of a bf16 boundary: rare, one bf16 ulp each
    e = ((emu - direct).abs().max(1)[0] / scale).numpy()
    assert np.median(e) < 2e-6 and (e > 1e-4).mean() < 0.02 and e.max() < 5e-3, (np.median(e), (e > 1e-4).mean(), e.max())
    vs_ref = float((emu - T(g[name + "_out"]).double()).abs().max() / scale)
    assert 1e-3 < vs_ref < 2e-2, vs_ref


def test_philox_known_answers():
    """oracle/philox.py against the known-answer vectors of Philox4x32-10 (Random123 kat_vectors)."""
    from oracle import philox
    for counter, key, expect in philox.KNOWN_ANSWERS:
        got = philox.philox4x32_10(counter, key)
        assert tuple(int(x) for x in got) == expect


def test_philox_table_layout_and_distribution():
    """The table the in-kernel draws come from: word w of block b is column 4 b + w, rows 2 r / 2 r + 1 are a ray's offsets /
    importance draws, a sub-table is a slice of the table (what lets chunks and row bands draw consistently), and the values
    are 24-bit uniforms on [0, 1) like torch.rand's."""
    from oracle import philox
    seed = 0x0123456789ABCDEF
    t = philox.uniform_table(seed, 0, 0, 64, 64)
    u = philox.uniform_table(seed, 1, 0, 64, 128)
    w = philox.philox4x32_10((3, 0, 2 * 5 + 1, 0), (seed & 0xFFFFFFFF, seed >> 32))
    assert u[5, 12:16].tolist() == [np.float32(int(x) >> 8) * np.float32(2.0 ** -24) for x in w]
    assert np.array_equal(philox.uniform_table(seed, 1, 40, 24, 128), u[40:])
    assert np.array_equal(philox.uniform_table(seed, 0, 0, 64, 61), t[:, :61])        # ragged widths: a prefix of the row
    assert not np.array_equal(t[:, :64], u[:, :64]) and not np.array_equal(philox.uniform_table(seed + 1, 0, 0, 64, 64), t)
    big = philox.uniform_table(7, 1, 1 << 33, 4096, 128)                               # rows beyond 32 bits
    assert big.dtype == np.float32 and big.min() >= 0.0 and big.max() < 1.0
    assert np.array_equal(big * 2 ** 24, np.floor(big * 2 ** 24))
    assert abs(float(big.mean()) - 0.5) < 2e-3 and abs(float(big.var()) - 1 / 12) < 1e-3
    assert len(np.unique(big)) > 0.98 * big.size
