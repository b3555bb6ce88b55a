"""The reference-NAMED callables of the drop-in surface, called the way a reference caller calls them
(NeRFs/HeadNeRF/helper.py:228-313, train/baseline.py:325-375, train/audio_exp_nerf.py:274-288,369-387),
on the reference-generated goldens and with the tolerances of the ops-level tests in
test_hip_parity.py.  Also the argument guards of the tensor layer (wrong device, wrong shape, wrong
dtype raise IdealNerfError before any C call).  Needs a real MI355X: run with ``-m gpu``.
"""
import numpy as np
import pytest
import torch

import oracle
from parity_proof import flipped_rows, oracle_fine_pass, prove_render

pytestmark = pytest.mark.gpu

NEAR, FAR = 0.5772005200386048, 1.1772005200386046
RGB_TOL, W_TOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def idn():
    import idealnerf_amd
    idealnerf_amd._lib.load()
    return idealnerf_amd


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def abs_err(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    return np.abs(a - b).max()


def scale_sigma(p, gain=300.0, bias=0.3):
    p = {k: v.clone() for k, v in p.items()}
    p["alpha_linear.weight"] = p["alpha_linear.weight"] * gain
    p["alpha_linear.bias"] = torch.full_like(p["alpha_linear.bias"], bias)
    return p


@pytest.fixture(scope="module")
def frame_net(idn, dev):
    """The Network of the reference's 32x32 golden frame (tests/golden/frame32.npz)."""
    from idealnerf_amd.audio_exp_nerf import Network
    from idealnerf_amd.helper import RenderConfig
    dims = oracle.facenerf_dims()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    cfg = RenderConfig(perturb=0.0, chunk=300, near=NEAR, far=FAR)
    net = Network(32, 32, syn["focal"], NEAR, FAR, 300, None, 64, 128, args=cfg).to(dev)
    net.face_nerf_coarse.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(2, dims)))
    net.face_nerf_fine.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(3, dims)))
    return net.eval(), syn


# --------------------------------------------------------------------------- helper.sample_pdf
def test_helper_sample_pdf_is_the_kernels_stage(idn, dev, golden):
    """helper.sample_pdf(bins, weights, N, det / pytest / u): bit-identical samples to the reference for the
    deterministic and the numpy-seeded draw, and the SAME cdf arithmetic as Network.render_rays' stage."""
    from idealnerf_amd import helper
    g = golden("sample_pdf")
    bins, w = T(g["bins"]).to(dev), T(g["weights"]).to(dev)
    det = helper.sample_pdf(bins, w, 128, det=True)
    np.testing.assert_array_equal(det.cpu().numpy(), g["det_samples"])
    rnd = helper.sample_pdf(bins, w, 128, det=False, pytest=True)      # np.random.seed(0); rand(n, N): helper.py:286-293
    np.testing.assert_array_equal(rnd.cpu().numpy(), g["rnd_samples"])
    given = helper.sample_pdf(bins, w, 128, u=T(g["rnd_u"]).to(dev))
    np.testing.assert_array_equal(given.cpu().numpy(), g["rnd_samples"])
    free = helper.sample_pdf(bins, w, 128)                                # torch.rand on the device: in range, finite
    assert free.shape == (64, 128) and bool(torch.isfinite(free).all())
    assert bool((free >= bins[:, :1]).all()) and bool((free <= bins[:, -1:]).all())
    # one stage, two entry points: the renderer's (z, weights[n, S]) form gives the same cdf and indices
    f = golden("frame32")
    z, wc = T(f["tap_z_coarse"]).to(dev), T(f["tap_weights_coarse"]).to(dev)
    mids = 0.5 * (z[:, 1:] + z[:, :-1])
    a = idn.ops.sample_pdf_bins_fwd(mids.contiguous(), wc[:, 1:-1].contiguous(), T(f["tap_u"])[0].contiguous().to(dev))
    b = idn.ops.sample_pdf_fwd(z, wc, T(f["tap_u"])[0].contiguous().to(dev), 128)
    for k in ("cdf", "inds", "z_samples"):
        assert torch.equal(a[k], b[k]), k
    np.testing.assert_array_equal(helper.sample_pdf(mids, wc[:, 1:-1], 128, det=True).cpu().numpy(), f["tap_z_samples"])


# --------------------------------------------------------------------------- helper.raw2outputs / Network.raw2outputs
@pytest.mark.parametrize("S", [64, 192])
def test_helper_raw2outputs_golden(idn, dev, golden, frame_net, S):
    from idealnerf_amd import helper
    g = golden("raw2outputs")
    k = lambda n: T(g[f"s{S}_{n}"]).to(dev)
    net, _ = frame_net
    for fn in (helper.raw2outputs, net.raw2outputs):
        rgb_map, disp_map, acc_map, weights, depth_map = fn(k("raw"), k("z"), k("d"), k("bc"))
        for got, name in ((rgb_map, "rgb_map"), (disp_map, "disp"), (acc_map, "acc"), (weights, "weights"),
                          (depth_map, "depth")):
            assert rel_err(got, g[f"s{S}_{name}"]) < 2e-6, name


# --------------------------------------------------------------------------- helper.get_rays
def test_helper_get_rays_golden(idn, dev, golden):
    from idealnerf_amd import helper
    g = golden("frame32")
    syn = oracle.synthetic_frame(32, 32, seed=0)
    ro, rd = helper.get_rays(32, 32, syn["focal"], syn["c2w"], device=dev)
    assert ro.shape == (32, 32, 3) and rd.shape == (32, 32, 3)
    assert rel_err(ro.reshape(-1, 3), g["rays"][:, 0:3]) < 1e-6
    assert rel_err(rd.reshape(-1, 3), g["rays"][:, 3:6]) < 1e-6
    ro_ref, rd_ref = oracle.camera_rays(32, 32, syn["focal"], syn["c2w"])
    assert rel_err(rd, rd_ref) < 1e-6 and rel_err(ro, ro_ref) < 1e-6


# --------------------------------------------------------------------------- Network.run_network
def test_network_run_network_golden(idn, dev, golden, frame_net):
    """run_network(pts, expr, viewdirs, aud, model, latent) -> raw: the reference's coarse and fine raw taps."""
    g = golden("frame32")
    net, syn = frame_net
    rays = T(g["rays"])[:128].to(dev)
    cond = dict(expr=syn["expr"].to(dev), aud=syn["aud"].to(dev), latent_code=syn["latent"].to(dev))
    for z_key, raw_key, model in (("tap_z_coarse", "tap_raw_coarse", net.face_nerf_coarse),
                                  ("tap_z_fine", "tap_raw_fine", net.face_nerf_fine)):
        z = T(g[z_key])[:128].to(dev)
        pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]          # audio_exp_nerf.py:331
        with torch.no_grad():
            raw = net.run_network(pts, cond["expr"], rays[:, 8:11].contiguous(), cond["aud"], model, cond["latent_code"])
        assert raw.shape == (128, z.shape[1], 4)
        assert rel_err(raw, g[raw_key]) < 2e-5, raw_key
    # gradients are not built on this entry: with autograd on it raises rather than returning a dead tensor
    z = T(g["tap_z_coarse"])[:128].to(dev)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    with pytest.raises(NotImplementedError):
        net.run_network(pts, cond["expr"], rays[:, 8:11].contiguous(), cond["aud"], net.face_nerf_coarse, cond["latent_code"])


# --------------------------------------------------------------------------- Network.batchify_rays / render_rays
def test_network_batchify_rays_golden(idn, dev, golden, frame_net):
    """batchify_rays at a ragged chunk (300 of 1024 rays): the reference frame, every key of the dict."""
    g = golden("frame32")
    net, syn = frame_net
    rays, bc = T(g["rays"]).to(dev), syn["bc"].reshape(-1, 3).to(dev)
    args = (syn["aud"].to(dev), syn["c2w"], syn["latent"].to(dev), syn["expr"].to(dev))
    with torch.no_grad():
        one = net.batchify_rays(rays, bc, *args, chunk=300)
        assert set(one) == {"rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std", "last_weight"}
        for k, gk in (("rgb_map", "rgb"), ("rgb0", "rgb0")):
            assert one[k].shape == (1024, 3) and rel_err(one[k], g[gk].reshape(-1, 3)) < RGB_TOL, k
        for k, gk in (("disp_map", "disp"), ("acc_map", "acc"), ("disp0", "disp0"), ("acc0", "acc0")):
            assert rel_err(one[k], g[gk].reshape(-1)) < RGB_TOL, k
        assert abs_err(one["last_weight"], g["last_weight"].reshape(-1)) < W_TOL
        # perturb > 0 (the reference's default, also in eval): one call as well, the frame's draws from torch's generator --
        # reproducible under a seed, stochastic otherwise, and close to the deterministic frame
        net.args.perturb = 1.0
        try:
            torch.manual_seed(0)
            jit_a = net.batchify_rays(rays, bc, *args, chunk=300)
            torch.manual_seed(0)
            jit_b = net.batchify_rays(rays, bc, *args, chunk=8192)
            jit_c = net.batchify_rays(rays, bc, *args, chunk=300)
        finally:
            net.args.perturb = 0.0
        assert jit_a["rgb_map"].shape == (1024, 3) and bool(torch.isfinite(jit_a["rgb_map"]).all())
        assert torch.equal(jit_a["rgb_map"], jit_b["rgb_map"]) and not torch.equal(jit_a["rgb_map"], jit_c["rgb_map"])
        assert float((jit_a["rgb_map"] - one["rgb_map"]).abs().mean()) < 0.02
        chunked = torch.cat([net.render_rays(rays[i:i + 300], bc[i:i + 300], *args)["rgb_map"] for i in range(0, 1024, 300)], 0)
        assert torch.equal(chunked, one["rgb_map"])


def test_network_render_rays_pytest_draws_golden(idn, dev, golden, frame_net):
    """render_rays(perturb=1, pytest=True): the reference's numpy-seeded t_rand / u (audio_exp_nerf.py:316-326,
    helper.py:286-293) drawn by the drop-in itself."""
    g, f = golden("rays64_jitter"), golden("frame32")
    net, syn = frame_net
    sel = T(g["sel"])
    with torch.no_grad():
        out = net.render_rays(T(f["rays"])[sel].to(dev), syn["bc"].reshape(-1, 3)[sel].to(dev), syn["aud"].to(dev),
                              syn["c2w"], syn["latent"].to(dev), syn["expr"].to(dev), perturb=1.0, pytest=True, taps=True)
    np.testing.assert_array_equal(out["tap_z_coarse"].cpu().numpy(), g["z_coarse"])
    for k in ("rgb_map", "rgb0", "disp_map", "acc_map"):
        assert rel_err(out[k], g[k]) < RGB_TOL, k
    assert abs_err(out["last_weight"], g["last_weight"]) < W_TOL


# --------------------------------------------------------------------------- packed-stream cache
def test_packed_stream_follows_weight_updates(idn, dev):
    """The cached MFMA weight stream is rebuilt after optimizer steps and load_state_dict; after a write
    through ``.data`` (which PyTorch's version counter does not see) invalidate_packed() rebuilds it."""
    dims = oracle.facenerf_dims()
    net = idn.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76).to(dev)
    net.load_state_dict(oracle.xavier_facenerf_params(5, dims))
    rs = np.random.RandomState(0)
    x = T(rs.uniform(-1, 1, size=(200, 90)).astype(np.float32)).to(dev)
    cond = [T(rs.standard_normal(k).astype(np.float32)).to(dev) for k in (64, 76, 32)]

    def check():
        with torch.no_grad():
            out = net(x, *cond)
            ref = oracle.facenerf_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, x.cpu(),
                                          *[c.cpu() for c in cond], dims)
        assert rel_err(out, ref) < 1e-5

    check()
    with torch.no_grad():
        net.pts_linears[3].weight.mul_(1.5)            # in-place op: version bump, cache key changes
    check()
    net.load_state_dict(oracle.xavier_facenerf_params(6, dims))
    check()
    net.pts_linears[2].weight.data.mul_(0.5)            # .data write: invisible to the version counter
    idn.invalidate_packed(net)
    check()
    net.rgb_linear.bias.data.fill_(0.25)                # biases are folded per call: no invalidation needed
    check()


# --------------------------------------------------------------------------- argument guards
def test_ops_reject_bad_shapes_devices_and_dtypes(idn, dev):
    E = idn._lib.IdealNerfError
    ops = idn.ops
    rays = torch.zeros((16, 11), device=dev)
    z = torch.linspace(0.6, 1.0, 64, device=dev).expand(16, 64).contiguous()
    raw = torch.zeros((16, 64, 4), device=dev)
    bc = torch.zeros((16, 3), device=dev)
    t = torch.linspace(0, 1, 64, device=dev)
    with pytest.raises(E, match=r"rays must be \[16, 11\]"):
        ops.composite_fwd(raw, z, rays[:, :8].contiguous(), bc)        # the reference's 8-column ray batch
    with pytest.raises(E, match="bc_rgb"):
        ops.composite_fwd(raw, z, rays, bc[:, :2].contiguous())
    with pytest.raises(E, match="raw"):
        ops.composite_fwd(raw[:, :63].contiguous(), z, rays, bc)
    with pytest.raises(E, match="rays"):
        ops.coarse_depths(rays[:, :8].contiguous(), t)
    with pytest.raises(E, match="t_rand"):
        ops.coarse_depths(rays, t, torch.zeros((16, 63), device=dev))
    with pytest.raises(E, match="weights"):
        ops.sample_pdf_fwd(z, z[:, :62].contiguous(), torch.linspace(0, 1, 128, device=dev), 128)
    with pytest.raises(E, match="u must be"):
        ops.sample_pdf_fwd(z, z, torch.zeros((15, 128), device=dev), 128)
    with pytest.raises(E, match="u must be"):
        ops.sample_pdf_fwd(z, z, torch.linspace(0, 1, 64, device=dev), 128)
    with pytest.raises(E, match="GPU"):
        ops.composite_fwd(raw.cpu(), z, rays, bc)
    with pytest.raises(E, match="float32"):
        ops.composite_fwd(raw.double(), z, rays, bc)
    with pytest.raises(E, match="contiguous"):
        ops.composite_fwd(raw, z.t().contiguous().t(), rays, bc)
    with pytest.raises(E, match="viewdirs"):
        ops.query_points_fwd(None, None, torch.zeros((4, 8, 3), device=dev), torch.zeros((5, 3), device=dev))
    dims = oracle.facenerf_dims()
    net = idn.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76).to(dev)
    packed = net.packed_weights("f32")     # (the ops default to the fp32 kernel whatever the modules' default arithmetic is)
    folded = net.folded_bias(*[torch.zeros(k, device=dev) for k in (64, 76, 32)])
    with pytest.raises(E, match="packed"):
        ops.query_rays_fwd(packed[:-4].contiguous(), folded, rays, z)
    with pytest.raises(E, match="folded"):
        ops.query_rays_fwd(packed, folded[:-1].contiguous(), rays, z)
    with pytest.raises(E, match="packed"):                              # a stream packed for another arithmetic
        ops.query_rays_fwd(net.packed_weights("bf16"), folded, rays, z, precision=0)
    with pytest.raises(E, match="rows"):
        ops.frame_rays(torch.eye(4), 32, 32, 100.0, NEAR, FAR, row0=30, nrows=5, device=dev)
    if torch.cuda.device_count() > 1:
        other = torch.device("cuda", 1)
        with pytest.raises(E, match="same GPU"):
            ops.composite_fwd(raw, z.to(other), rays, bc)


def test_launch_follows_the_tensors_device_and_stream(idn, dev):
    """The launch goes to the tensors' device on that device's current stream: work queued on a side stream
    is ordered with that stream's other work and not with the default stream's."""
    ops = idn.ops
    rs = np.random.RandomState(1)
    n, S = 4096, 64
    raw = T(rs.standard_normal((n, S, 4)).astype(np.float32)).to(dev)
    z = torch.sort(T(rs.uniform(NEAR, FAR, (n, S)).astype(np.float32)), -1)[0].to(dev)
    rays = torch.zeros((n, 11), device=dev)
    rays[:, 3:6] = T(rs.standard_normal((n, 3)).astype(np.float32)).to(dev)
    bc = T(rs.uniform(0, 1, (n, 3)).astype(np.float32)).to(dev)
    ref = ops.composite_fwd(raw, z, rays, bc)["rgb_map"].clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        raw2 = raw * 1.0                                   # produced on the side stream ...
        out = ops.composite_fwd(raw2, z, rays, bc)["rgb_map"]   # ... and consumed there without a sync in between
    side.synchronize()
    assert torch.equal(out, ref)


# --------------------------------------------------------------------------- lindisp / white_bkgd / raw_noise_std
def test_render_switches_golden(idn, dev, golden, frame_net):
    """The three switches the reference's Network leaves at their defaults (audio_exp_nerf.py:297-299): through
    helper.raw2outputs and through Network.render_rays, against the reference's own outputs."""
    from idealnerf_amd import helper
    g, r, f = golden("flags"), golden("raw2outputs"), golden("frame32")
    k = lambda n: T(r["s64_" + n]).to(dev)
    for tag, kw in (("white", dict(white_bkgd=True)), ("noise", dict(raw_noise_std=0.7, pytest=True)),
                    ("both", dict(raw_noise_std=2.5, white_bkgd=True, pytest=True))):
        out = helper.raw2outputs(k("raw"), k("z"), k("d"), k("bc"), **kw)
        for got, name in zip(out, ("rgb_map", "disp", "acc", "weights", "depth")):
            assert rel_err(got, g[f"r2o_{tag}_{name}"]) < 2e-6, (tag, name)
    net, syn = frame_net
    sel = T(g["sel"])
    rays, bc = T(f["rays"])[sel].to(dev), syn["bc"].reshape(-1, 3)[sel].to(dev)
    args = (syn["aud"].to(dev), syn["c2w"], syn["latent"].to(dev), syn["expr"].to(dev))
    with torch.no_grad():
        for tag, kw in (("lindisp", dict(lindisp=True)), ("white", dict(white_bkgd=True)),
                        ("noise", dict(raw_noise_std=0.5, pytest=True, perturb=0.0))):
            out = net.render_rays(rays, bc, *args, taps=True, **kw)
            for key in ("rgb_map", "rgb0", "disp_map", "acc_map", "disp0", "acc0"):
                assert rel_err(out[key], g[f"rr_{tag}_{key}"]) < RGB_TOL, (tag, key)
            assert abs_err(out["last_weight"], g[f"rr_{tag}_last_weight"]) < W_TOL
            assert rel_err(out["tap_z_coarse"], g[f"rr_{tag}_z_coarse"]) < 2e-7, tag
    np.testing.assert_array_equal(net.render_rays(rays, bc, *args, taps=True)["tap_z_coarse"].cpu().numpy(),
                                  g["rr_white_z_coarse"])
    # the training path takes lindisp (depths carry no gradient) and refuses the two compositing switches
    net.train()
    try:
        with pytest.raises(NotImplementedError):
            net.render_rays(rays, bc, *args, white_bkgd=True)
        out = net.render_rays(rays, bc, *args, lindisp=True, perturb=0.0)
        assert rel_err(out["rgb_map"], g["rr_lindisp_rgb_map"]) < RGB_TOL and out["rgb_map"].requires_grad
    finally:
        net.eval()


# --------------------------------------------------------------------------- audio nets on the GPU
def test_audio_nets_on_gpu_match_reference_and_conv_autograd(idn, dev, golden):
    """On the GPU the kernel-3 convolutions of AudioNet / AudioAttNet run as a window gather + one matrix
    product (MIOpen's naive conv kernels stay off the training step): forward against the reference's outputs,
    gradients against nn.Conv1d's own autograd on the CPU, and the batched clip pass against the per-frame loop."""
    from idealnerf_amd.models.audio_net import AudioAttNet, AudioNet, DeepSpeechAudNet, clip_audio_features
    g = golden("audio_nets")
    nets = {"aud": AudioNet(64, 16), "att": AudioAttNet(), "ds": DeepSpeechAudNet()}
    for tag, m in nets.items():
        m.load_state_dict({k[len(tag) + 4:]: T(v) for k, v in g.items() if k.startswith(tag + "_sd_")}, strict=True)
    cpu_nets = {k: type(m)(*((64, 16) if k == "aud" else ())) for k, m in nets.items()}
    for k in nets:
        cpu_nets[k].load_state_dict(nets[k].state_dict())
        nets[k].to(dev)
    auds = T(g["aud_in"]).to(dev)
    with torch.no_grad():
        out8 = nets["aud"](auds)
        np.testing.assert_allclose(out8.cpu().numpy(), g["aud_out8"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(nets["aud"](auds[3:4]).cpu().numpy(), g["aud_out1"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(nets["att"](out8).cpu().numpy(), g["att_out"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(nets["ds"](auds[3:4]).cpu().numpy(), g["ds_out"], rtol=2e-5, atol=2e-6)
    # gradients of a scalar through AudioNet -> AudioAttNet: GPU (matmul form) vs CPU (nn.Conv1d)
    w = T(np.random.RandomState(3).standard_normal(64).astype(np.float32))
    loss_gpu = (nets["att"](nets["aud"](auds)) * w.to(dev)).sum()
    loss_cpu = (cpu_nets["att"](cpu_nets["aud"](auds.cpu())) * w).sum()
    loss_gpu.backward()
    loss_cpu.backward()
    assert abs(float(loss_gpu) - float(loss_cpu)) < 1e-5 * max(1.0, abs(float(loss_cpu)))
    for k in ("aud", "att"):
        for (name, p), (_, q) in zip(nets[k].named_parameters(), cpu_nets[k].named_parameters()):
            assert rel_err(p.grad, q.grad) < 2e-4, (k, name)
    c = golden("audio_clip")
    aud_net, att_net = AudioNet(64, 16), AudioAttNet()
    aud_net.load_state_dict({k[len("audnet."):]: T(v) for k, v in c.items() if k.startswith("audnet.")})
    att_net.load_state_dict({k[len("attnet."):]: T(v) for k, v in c.items() if k.startswith("attnet.")})
    with torch.no_grad():
        feats = clip_audio_features(aud_net.to(dev), att_net.to(dev), T(c["auds"]).to(dev))
    np.testing.assert_allclose(feats.cpu().numpy(), c["out"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("n,dim_aud", [(1, 64), (8, 64), (3, 76), (40, 64)])
def test_fused_audio_net_matches_the_eager_module(idn, dev, n, dim_aud):
    """AudioNet (models/audio_net.py:43-69) on the GPU is one HIP kernel forward and one backward (csrc/audio.hip); the eager
    PyTorch module with the same parameters is the yardstick: outputs to 2e-6, every parameter gradient to 1e-5 of its largest
    entry, for one window (a frame / a training step), the smoother's eight, and -- forward only -- a whole clip."""
    from idealnerf_amd.models import audio_net as AN
    torch.manual_seed(n)
    net = AN.AudioNet(dim_aud, 16).to(dev)
    x = torch.randn(n, 16, 29, device=dev)
    w = torch.randn(n, dim_aud, device=dev).squeeze()

    def run(fused):
        old, AN.FUSED_AUDIO_NET = AN.FUSED_AUDIO_NET, fused
        try:
            for p_ in net.parameters():
                p_.grad = None
            if n > 8:
                with torch.no_grad():
                    return net(x), None
            out = net(x)
            (out * w).sum().backward()
            return out.detach(), [p_.grad.clone() for p_ in net.parameters()]
        finally:
            AN.FUSED_AUDIO_NET = old

    out_f, g_f = run(True)
    out_e, g_e = run(False)
    assert out_f.shape == out_e.shape == ((dim_aud,) if n == 1 else (n, dim_aud))
    assert rel_err(out_f, out_e) < 2e-6
    if g_f is not None:
        for (name, _), a, b in zip(net.named_parameters(), g_f, g_e):
            assert rel_err(a, b) < 1e-5, name
        out2, g2 = run(True)     # deterministic: no atomics
        assert torch.equal(out2, out_f) and all(torch.equal(a, b) for a, b in zip(g2, g_f))
    else:   # more than eight windows with gradients fall back to the eager module (the backward kernel holds eight windows in LDS)
        net(x).sum().backward()
        assert all(p_.grad is not None for p_ in net.parameters())


# --------------------------------------------------------------------------- BASELINE configs[3]: Obama, by name
@pytest.mark.parametrize("cfg_name", ["NeRFs/HeadNeRF/configs/audio_expr_nerf/obama/paper_model.txt",
                                      "NeRFs/HeadNeRF/configs/audio_expr_nerf/obama3/paper_model/torso_bg.txt",
                                      "NeRFs/HeadNeRF/configs/audio_expr_nerf/obama3/paper_model_aud_only.txt"])
def test_obama_configs_render_by_name(idn, dev, cfg_name):
    """The reference's shipped Obama configs (their text is part of tests/golden/configs_parsed.json), parsed by
    the product's config layer into the Network: near/far 0.567-0.634 / 1.167-1.234 and dim_expr 79 or 0 instead
    of May's constants; a row band of the frame against the CPU oracle (the multi-GPU partition of configs[3])."""
    import json, os
    from idealnerf_amd import config
    from idealnerf_amd.audio_exp_nerf import Network
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "configs_parsed.json")))[cfg_name]
    ns = config.load_config(text="\n".join(gold["lines"]))
    cfg = config.to_render_config(ns)
    cfg.perturb = 0.0
    assert (cfg.near, cfg.far) == (gold["parsed"]["near"], gold["parsed"]["far"]) and cfg.near != NEAR
    H = W = 24
    dims = oracle.facenerf_dims(dim_aud=cfg.dim_aud, dim_expr=cfg.dim_expr, dim_latent=cfg.dim_latent)
    syn = oracle.synthetic_frame(H, W, seed=3, dims=dims)
    net = Network(H, W, syn["focal"], cfg.near, cfg.far, cfg.chunk, None, cfg.N_samples, cfg.N_importance, args=cfg).to(dev).eval()
    pc, pf = scale_sigma(oracle.xavier_facenerf_params(31, dims)), scale_sigma(oracle.xavier_facenerf_params(32, dims))
    net.face_nerf_coarse.load_state_dict(pc)
    net.face_nerf_fine.load_state_dict(pf)
    g = lambda t: None if t is None else t.to(dev)
    rows = (8, 14)     # rank 1's band of a 3-way split
    expr = g(syn["expr"]) if cfg.dim_expr else None
    with torch.no_grad():
        rgb, disp, acc, last_w, extras = net.render_dynamic_face(
            H, W, syn["focal"], expr=expr, poses=syn["c2w"], latent_code=g(syn["latent"]),
            render_poses=syn["c2w"][:3, :4], chunk=cfg.chunk, near=cfg.near, far=cfg.far, bc_rgb=g(syn["bc"]),
            aud_para=g(syn["aud"]), rows=rows)
        ref = oracle.render_frame(H, W, syn["focal"], syn["c2w"], cfg.near, cfg.far, syn["bc"], pc, pf, syn["aud"],
                                  syn["expr"] if cfg.dim_expr else None, syn["latent"], dims=dims, rows=rows, taps=True)
        # the band once more with the debug taps, by the call render_dynamic_face makes (same rays, same kernels)
        rays = idn.ops.frame_rays(syn["c2w"][:3, :4], H, W, syn["focal"], cfg.near, cfg.far, rows[0], rows[1] - rows[0], device=dev)
        bc = g(syn["bc"])[rows[0]:rows[1]].reshape(-1, 3).contiguous()
        tapped = net.render_rays(rays, bc, g(syn["aud"]), syn["c2w"], g(syn["latent"]), expr, taps=True)
        ff = net.face_nerf_fine.folded_bias(g(syn["aud"]), expr, g(syn["latent"]))
    assert rgb.shape == (6, W, 3) and torch.equal(tapped["rgb_map"], rgb.reshape(-1, 3))
    flat = {k: v.reshape((-1,) + tuple(v.shape[2:])) for k, v in ref.items()}
    # fixed 1e-4 behind the sampling, exact sampling stage, coarse weights within 1e-5: every ray (tests/parity_proof.py)
    prove_render(idn, cfg_name.split("/")[-2] + " band", tapped, flat, net.face_nerf_fine.packed_weights(), ff, rays, bc,
                 lambda z: oracle_fine_pass(pf, dims, rays, bc, syn["aud"], syn["expr"] if cfg.dim_expr else None, syn["latent"], z), 2e-4,
                 precision_fine=net.face_nerf_fine.prec_code)
    fl, _ = flipped_rows(tapped["tap_inds"], flat["tap_inds"])
    keep = torch.from_numpy(~fl)
    # last_weight is bounded by 1 and may be ~1e-17: absolute error, on the rays whose sample positions are the oracle's
    assert abs_err(last_w.reshape(-1)[keep.to(dev)], flat["last_weight"][keep]) < RGB_TOL
    assert rel_err(extras["rgb0"], ref["rgb0"]) < RGB_TOL


def test_forward_smoother_golden(idn, dev, golden):
    """The reference's `Network.forward` behind `nosmo_iters` (audio_exp_nerf.py:228-279; tests/golden/smoother.npz): eight-frame
    audio window with zero padding at the clip's ends -> AudioNet (one HIP kernel for the eight windows) -> AudioAttNet ->
    full-frame render.  The audio feature handed to the renderer and the rendered 12 x 12 frame against the reference's, for a
    frame at the start, in the middle and at the end of the clip."""
    from test_boundary_cpu import _smoother_data, _smoother_network
    g = golden("smoother")
    net = _smoother_network(idn, g, dev)
    dims = oracle.facenerf_dims()
    net.face_nerf_coarse.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(2, dims), 300.0, 0.3))
    net.face_nerf_fine.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(3, dims), 300.0, 0.3))
    real = net.render_dynamic_face
    seen = {}

    def spy(*a, **k):
        seen["aud"] = k["aud_para"].detach().clone()
        return real(*a, **k)

    net.render_dynamic_face = spy
    for idx in g["frames"]:
        with torch.no_grad():
            rgb, disp, acc, last_w, extras = net([_smoother_data(g, int(idx)), int(g["nosmo_iters"]), 10])
        np.testing.assert_allclose(seen["aud"].cpu().numpy(), g[f"aud_feature_{int(idx)}"], rtol=2e-5, atol=2e-6, err_msg=f"frame {idx}")
        assert rgb.shape == (12, 12, 3)
        assert rel_err(extras["rgb0"], g[f"rgb0_{int(idx)}"]) < RGB_TOL, idx     # nothing is sampled before the coarse composite
        e = np.abs(rgb.cpu().numpy().astype(np.float64) - g[f"rgb_{int(idx)}"]).reshape(-1, 3).max(1) / np.abs(g[f"rgb_{int(idx)}"]).max()
        print(f"\n  smoother frame {int(idx)}: rgb vs the reference max {e.max():.2e}, {int((e > RGB_TOL).sum())} of {e.size} rays beyond 1e-4")
        assert e.max() < 1e-3 and (e > RGB_TOL).sum() <= 3     # (the end-to-end guards of tests/parity_proof.py; measured: 0 rays)


def test_forward_dim_aud_29_golden(idn, dev, golden):
    """The reference's `dim_aud = 29` configuration (audio_exp_nerf.py:265-269: `ds_aud_net`, a Linear(16, 1) squeeze of the
    DeepSpeech window, instead of AudioNet; FaceNeRFs with 29 audio columns, C = 200) through `Network.forward` in eval mode:
    tests/golden/ds29.npz is the reference's own forward on a 12 x 12 frame.  A fourth conditioning layout of the same kernels."""
    from idealnerf_amd.audio_exp_nerf import Network
    from idealnerf_amd.helper import RenderConfig
    g = golden("ds29")
    net = Network(12, 12, float(g["focal"]), NEAR, FAR, 512, None, 64, 128,
                  args=RenderConfig(perturb=0.0, chunk=512, near=NEAR, far=FAR, dim_aud=29)).to(dev).eval()
    assert tuple(net.face_nerf_coarse.pts_linears[0].weight.shape) == (256, 200)
    dims = oracle.facenerf_dims(dim_aud=29)
    net.face_nerf_coarse.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(2, dims), 300.0, 0.3))
    net.face_nerf_fine.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(3, dims), 300.0, 0.3))
    net.ds_aud_net.load_state_dict({k[len("dsnet."):]: T(v).to(dev) for k, v in g.items() if k.startswith("dsnet.")})
    real, seen = net.render_dynamic_face, {}

    def spy(*a, **k):
        seen["aud"] = k["aud_para"].detach().clone()
        return real(*a, **k)

    net.render_dynamic_face = spy
    data = (torch.zeros(1, 2, 1, 3), torch.zeros(1, 3), T(g["bg"])[None], T(g["auds"])[None], torch.zeros(1, 12, 12, 3), T(g["pose"])[None],
            T(g["expr"])[None], T(g["latent"]), torch.tensor([int(g["index"])]))
    with torch.no_grad():
        rgb, disp, acc, last_w, extras = net([data, 0, 6])
    np.testing.assert_allclose(seen["aud"].cpu().numpy(), g["aud_feature"], rtol=1e-5, atol=1e-6)
    assert rel_err(extras["rgb0"], g["rgb0"]) < RGB_TOL
    e = np.abs(rgb.cpu().numpy().astype(np.float64) - g["rgb"]).reshape(-1, 3).max(1) / np.abs(g["rgb"]).max()
    print(f"\n  dim_aud 29: rgb vs the reference max {e.max():.2e}, {int((e > RGB_TOL).sum())} of {e.size} rays beyond 1e-4")
    assert e.max() < 1e-3 and (e > RGB_TOL).sum() <= 3


def test_network_under_dataparallel_like_the_eval_script(idn, dev):
    """The reference's eval script wraps the renderer in `nn.DataParallel` (test/eval_aud_exp_nerf.py:475; batch dimension 1, so
    one device does the work) and calls `network([data, global_step, dataset_size])` with the loader's CPU tensors: scatter moves
    them to the device, the module runs once.  The wrapped call must give the direct call's frame bit for bit."""
    from idealnerf_amd.audio_exp_nerf import Network
    from idealnerf_amd.helper import RenderConfig
    dims = oracle.facenerf_dims()
    syn = oracle.synthetic_frame(24, 24, seed=2, dims=dims)
    torch.manual_seed(5)
    net = Network(24, 24, syn["focal"], NEAR, FAR, 8192, None, 64, 128, args=RenderConfig(perturb=0.0, near=NEAR, far=FAR)).to(dev).eval()
    net.face_nerf_coarse.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(41, dims)))
    net.face_nerf_fine.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(42, dims)))
    auds = torch.randn(1, 8, 16, 29)
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)[None]
    data = (torch.zeros(1, 2, 1, 3), torch.zeros(1, 3), syn["bc"][None], auds, torch.zeros(1, 24, 24, 3), pose, syn["expr"][None],
            syn["latent"], torch.tensor([3]))
    with torch.no_grad():
        direct = net([data, 0, 8])
        wrapped = torch.nn.DataParallel(net, device_ids=[dev.index or 0])([data, 0, 8])
    assert direct[0].shape == (24, 24, 3) and float((direct[0] - syn["bc"].to(dev)).abs().mean()) > 0.02
    for a, b in zip(direct[:4], wrapped[:4]):
        assert torch.equal(a, b)
    assert set(direct[4]) == set(wrapped[4]) and all(torch.equal(direct[4][k], wrapped[4][k]) for k in direct[4])


# --------------------------------------------------------------------------- checkpoints on the GPU path
def _render_vs_oracle_on_its_state_dict(idn, dev, net, latent, what, n_rays=256, seed=5):
    """`n_rays` rays of a 32 x 32 frame through `net` (whatever weights it holds NOW), held to the CPU oracle evaluated on
    the SAME state dict with parity_proof's fixed budgets (coarse weights 1e-5, sampling stage exact, fine pass 1e-4)."""
    dims = oracle.facenerf_dims(dim_aud=net.args.dim_aud, dim_expr=net.args.dim_expr, dim_latent=net.args.dim_latent)
    syn = oracle.synthetic_frame(32, 32, seed=seed, dims=dims)
    g = lambda t: None if t is None else t.to(dev)
    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pc, pf = cpu(net.face_nerf_coarse), cpu(net.face_nerf_fine)
    rays = idn.ops.frame_rays(syn["c2w"][:3, :4], 32, 32, syn["focal"], NEAR, FAR, device=dev)[:n_rays].contiguous()
    bc = g(syn["bc"]).reshape(-1, 3)[:n_rays].contiguous()
    lat = latent.detach().cpu()
    expr = syn["expr"] if net.args.dim_expr else None
    with torch.no_grad():
        out = net.render_rays(rays, bc, g(syn["aud"]), syn["c2w"], g(lat), g(expr), taps=True)
        ff = net.face_nerf_fine.folded_bias(g(syn["aud"]), g(expr), g(lat))
        ref = oracle.render_rays(rays.cpu(), bc.cpu(), pc, pf, syn["aud"], expr, lat, n_samples=64, n_importance=128, dims=dims, taps=True)
    assert float((ref["rgb_map"] - bc.cpu()).abs().mean()) > 0.02, "the volume must hide part of the background"
    prove_render(idn, what, out, ref, net.face_nerf_fine.packed_weights(), ff, rays, bc,
                 lambda z: oracle_fine_pass(pf, dims, rays, bc, syn["aud"], expr, lat, z), 1e-3,
                 precision_fine=net.face_nerf_fine.prec_code)
    assert rel_err(out["rgb0"], ref["rgb0"]) < RGB_TOL
    return out


def _head_network(idn, dev, seed, **cfg):
    from idealnerf_amd.audio_exp_nerf import Network, init_weights
    from idealnerf_amd.helper import RenderConfig
    torch.manual_seed(seed)
    net = Network(32, 32, 100.0, NEAR, FAR, 512, None, 64, 128, args=RenderConfig(perturb=0.0, chunk=512, near=NEAR, far=FAR, **cfg))
    net.apply(init_weights)                         # audio_exp_nerf.py:485
    idn.invalidate_packed(net)
    return net.to(dev).eval()


def _make_visible(net, gain=200.0, bias=0.3):
    """A trained head hides part of the background; Xavier initialisation does not (sigma ~ 0): scale the density heads."""
    with torch.no_grad():
        for m in (net.face_nerf_coarse, net.face_nerf_fine):
            m.alpha_linear.weight.mul_(gain)
            m.alpha_linear.bias.fill_(bias)


def test_head_tar_round_trip_renders_like_the_oracle(idn, dev, tmp_path):
    """The reference's resume path (audio_exp_nerf.py:516-525) and its writer (:586-591) on the GPU: a Network with
    non-trivial weights, a stepped Adam and per-frame latent codes is written as `head.tar` with the reference's four
    keys, found again by natural order, loaded into a FRESH Network (other weights, stale packed streams), and the loaded
    network's render is held to the CPU oracle on the checkpoint's own state dict and latent code -- and equals the
    saving network's render bit for bit."""
    from idealnerf_amd import checkpoint, train as T_
    net = _head_network(idn, dev, seed=1)
    _make_visible(net)
    lat = (1.0 + 0.1 * torch.randn(5, 32, generator=torch.Generator().manual_seed(2))).to(dev).requires_grad_(True)
    opt = T_.make_optimizer(net, lat)
    for p in list(net.parameters()) + [lat]:        # one real optimizer step so that the Adam state is non-trivial
        p.grad = 1e-3 * torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to(dev)
    opt.step()
    net.eval()
    run = tmp_path / "logs" / "may"
    checkpoint.save_checkpoint(str(run / "head.tar"), net, opt, lat, 4321)
    ck = torch.load(checkpoint.latest_checkpoint(str(run)), weights_only=False)
    assert set(ck) == {"global_step", "model_state_dict", "optimizer", "latent_codes"}        # :586-591
    assert set(ck["model_state_dict"]) == set(net.state_dict()) and "face_nerf_fine.feature_linear.weight" in ck["model_state_dict"]
    saved = _render_vs_oracle_on_its_state_dict(idn, dev, net, lat[3], "saving network")

    fresh = _head_network(idn, dev, seed=77)
    with torch.no_grad():                            # warm its packed streams with the WRONG weights first
        fresh.face_nerf_fine.packed_weights()
        fresh.face_nerf_coarse.packed_weights()
    lat2 = torch.ones(5, 32, device=dev, requires_grad=True)
    opt2 = T_.make_optimizer(fresh, lat2)
    step, codes = checkpoint.load_checkpoint(checkpoint.latest_checkpoint(str(run)), fresh, opt2, map_location=dev)
    lat2.data = codes                                # :523
    assert step == 4321 and torch.equal(codes, lat.data)
    for (k, a), (_, b) in zip(net.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k
    st, st2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert st.keys() == st2.keys() and all(torch.equal(st[i]["exp_avg_sq"], st2[i]["exp_avg_sq"]) for i in st)
    loaded = _render_vs_oracle_on_its_state_dict(idn, dev, fresh, lat2[3], "network loaded from head.tar")
    for k in ("rgb_map", "rgb0", "disp_map", "last_weight", "tap_inds", "tap_z_fine"):
        assert torch.equal(saved[k], loaded[k]), k


def test_adnerf_ft_path_warm_start_renders_like_the_oracle(idn, dev):
    """`--ft_path` (audio_exp_nerf.py:498-514): an AD-NeRF checkpoint -- audio-only FaceNeRFs (C = 127: input widths 127 /
    383 / 283) under `network_fn_state_dict` / `network_fine_state_dict`, plus the two audio nets -- warm-starts the paper
    model (C = 235): the three layers whose input width differs are dropped from the dict, everything else is loaded with
    strict=False.  Asserted: exactly which tensors changed, and that the render of the resulting weights matches the CPU
    oracle on them (the packed streams follow the load)."""
    from idealnerf_amd import checkpoint
    from idealnerf_amd.models.audio_net import AudioAttNet, AudioNet
    torch.manual_seed(4)     # (a Xavier density head is positive or negative over the whole volume: seed 3 draws an empty fine volume)
    ad_c, ad_f = idn.FaceNeRF(dim_aud=64, dim_latent=0, dim_expr=0), idn.FaceNeRF(dim_aud=64, dim_latent=0, dim_expr=0)
    from idealnerf_amd.audio_exp_nerf import init_weights
    for m in (ad_c, ad_f):
        m.apply(init_weights)
        with torch.no_grad():
            m.alpha_linear.weight.mul_(200.0)
            m.alpha_linear.bias.fill_(0.3)
    assert ad_c.pts_linears[0].weight.shape == (256, 127) and ad_c.pts_linears[5].weight.shape == (256, 383)
    aud_net, att_net = AudioNet(64, 16), AudioAttNet()
    ft = {"network_fn_state_dict": ad_c.state_dict(), "network_fine_state_dict": ad_f.state_dict(),
          "network_audnet_state_dict": aud_net.state_dict(), "network_audattnet_state_dict": att_net.state_dict(),
          "global_step": 400000}
    net = _head_network(idn, dev, seed=9)
    with torch.no_grad():
        net.face_nerf_fine.packed_weights()          # streams of the initialisation: must not survive the load
    before = {k: v.clone() for k, v in net.state_dict().items()}
    checkpoint.load_adnerf_finetune({k: (dict(v) if isinstance(v, dict) else v) for k, v in ft.items()}, net)
    after = net.state_dict()
    dropped = {"pts_linears.0.weight", "pts_linears.5.weight", "views_linears.0.weight"}
    for pre, src in (("face_nerf_coarse.", ad_c), ("face_nerf_fine.", ad_f)):
        for k, v in src.state_dict().items():
            if k in dropped:
                assert torch.equal(after[pre + k], before[pre + k]), f"{pre + k} has another input width: kept as initialised"
                assert after[pre + k].shape != v.shape
            else:
                assert torch.equal(after[pre + k].cpu(), v), pre + k
    for pre, src in (("aud_net.", aud_net), ("aud_att_net.", att_net)):
        for k, v in src.state_dict().items():
            assert torch.equal(after[pre + k].cpu(), v), pre + k
    changed = {k for k in after if not torch.equal(after[k], before[k])}
    assert not any(k.startswith("ds_aud_net.") for k in changed)
    _render_vs_oracle_on_its_state_dict(idn, dev, net, torch.ones(32), "network warm-started from an AD-NeRF ft_path")


# --------------------------------------------------------------------------- the driver's multi-GPU command, rehearsed
def test_bench_two_ranks_from_a_plain_start(dev):
    """`python bench.py --gpus 2 ...` as ONE plain process on this box: the launcher starts two ranks (sharing GPU 0
    over gloo -- the one-GPU rehearsal of the RCCL path), each renders its row band, the tiles are all-gathered,
    rank 0's line comes back.  Same kernels, same partition and the same code path as the 8-GPU run except for
    the transport."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (a rank that never reaches the rendezvous is an error after 90 s, not a silent wait)
    env = dict(os.environ, IDN_DIST_BACKEND="gloo", IDN_FORCE_DEVICE="0", IDN_DIST_TIMEOUT_S="90")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "96"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"] == 2 and res["backend"] == "gloo" and res["scaling"] == "strong"
    assert res["config"]["band_rows"] == [48, 48] and res["steps"] == 2
    assert res["value"] > 0 and res["roofline"]["frac"] is not None and res["roofline"]["traffic"] is None
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "96",
                          "--no-cpu-baseline", "--no-side-mode"], env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-3000:]
    r1 = json.loads([ln for ln in one.stdout.splitlines() if ln.strip()][-1])
    assert r1["n_gpus"] == 1 and r1["ranks"] == 1 and r1["config"]["rays_per_step"] == res["config"]["rays_per_step"]


def _two_rank_env():
    import os
    env = dict(os.environ, IDN_DIST_BACKEND="gloo", IDN_FORCE_DEVICE="0", IDN_DIST_TIMEOUT_S="90", OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_data_parallel_train_step_two_ranks(dev):
    """SURVEY 8(e), training: two ranks (sharing GPU 0 over gloo: the one-GPU rehearsal of the RCCL path) each render and
    back-propagate HALF of a 512-ray batch, `train_step` averages the gradients with one bucketed all-reduce and both take the
    same Adam step.  Asserted (tests/dp_train_worker.py): the replicas end with BIT-IDENTICAL parameters, and gradients and
    parameters equal a single-rank step on the concatenated batch to 1e-6 (the mean of the two half-batch gradients is the
    full-batch gradient; the dW GEMMs sum the points in another grouping, nothing else differs)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "dp_train_worker.py")],
                       env=_two_rank_env(), capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    print("\n  " + json.dumps(res))
    assert res["ranks"] == 2 and res["replicas_bit_identical"] is True
    assert res["params_moved_by"] > 1e-4, "the step must have moved the parameters"
    assert res["grad_rel_err_max"] < 1e-5, res                  # per tensor, relative to its largest entry (measured ~1e-6)
    assert res["param_abs_err_max"] < 1e-6, res
    assert abs(res["loss_single"] - res["loss_mean_of_ranks"]) < 1e-5 * max(1.0, abs(res["loss_single"]))


def test_bench_train_two_ranks_from_a_plain_start(dev):
    """`python bench.py --workload train --gpus 2` (round 3 refused it): the launcher's ranks train data-parallel behind
    `train.train_step -> parallel.average_gradients`, each on its own N_rand = 3072 rays, and end every step as one model."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "train", "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=_two_rank_env(), capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert res["n_gpus"] == 2 and res["ranks"] == 2 and res["backend"] == "gloo" and res["scaling"] == "weak"
    assert res["replicas_in_sync"] is True and res["value"] > 0 and np.isfinite(res["final_loss"])


def test_bench_one_rank_rccl_communicator(dev):
    """The RCCL branch of bench.py on the one GPU this box has: a ONE-rank "nccl" process group (device_id = this GPU,
    HSA_ENABLE_IPC_MODE_LEGACY=0 as the multi-GPU launcher sets it) and the frame's `all_gather_into_tensor` issued through
    it at world size 1 -- so backend initialisation, communicator creation and the collective call of the 8-GPU path
    (BASELINE configs[3]; NeRFs/HeadNeRF/train/distribute_nerf.py:457-466 upstream) have run at least once."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               IDN_DIST_INIT_WORLD1="1", IDN_DIST_TIMEOUT_S="90", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("IDN_DIST_BACKEND", None)
    p = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "96",
                          "--no-cpu-baseline", "--no-side-mode"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(p.pid, signal.SIGKILL)
        raise
    assert p.returncode == 0, err[-3000:]
    res = json.loads([ln for ln in out.splitlines() if ln.strip()][-1])
    assert res["backend"] == "nccl" and res["ranks"] == 1 and res["n_gpus"] == 1
    pr = res["per_rank"]
    assert len(pr["render_ms"]) == 1 and pr["render_ms"][0] > 0 and pr["all_gather_ms"][0] > 0      # the collective ran and was timed
    assert pr["step_ms_max"] <= res["ms_per_step"] * 1.05 + 1.0 and res["value"] > 0


def test_bench_refuses_more_ranks_than_gpus(dev):
    """`--gpus 2` over RCCL on a one-GPU node: every rank says what is wrong before any communicator is built."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    for k in ("IDN_DIST_BACKEND", "IDN_FORCE_DEVICE"):
        env.pop(k, None)
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU node")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "exposes 1 GPU(s)" in p.stderr
