"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference-generated golden vectors.  Needs a real MI355X: run with ``-m gpu``.

Tolerances (BASELINE.json north_star): RGB within 1e-4 relative; importance-sample
indices bit-exact at the searchsorted boundary (identical cdf/u in => identical inds
out); everything upstream of that boundary is fp32 arithmetic whose roundings differ
between a CPU BLAS and MFMA by ~1e-6, so end-to-end index agreement is reported as a
flip rate and bounded, not promised to be zero (hard part 1).
"""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

NEAR, FAR = 0.5772005200386048, 1.1772005200386046
RGB_TOL = 1e-4  # north_star: 1e-4 rel on RGB
BF16X6_CODE = 4  # IDN_PREC_BF16X6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def idn():
    import idealnerf_amd
    idealnerf_amd._lib.load()  # fail loudly if the HIP library is missing
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


# Comparisons with an oracle evaluated at run time use the FIXED 1e-4 -- and where a scene is sharp enough that the
# importance sampling turns a last-ulp difference of a coarse weight into more than that, the test proves exactly this
# and nothing more, for every ray (tests/parity_proof.py): coarse weights within 1e-5, the sampling stage bit-exact on
# the HIP weights, and the fine pass within 1e-4 of the oracle's on the oracle's AND on the HIP sample positions.
from parity_proof import (assert_default_precision_allowed, check_stage, default_precision_criterion, flipped_rows,  # noqa: E402
                          hip_fine_pass, oracle_fine_pass, prove, prove_render, small_sample_bound)


# Quantities bounded by 1 (weights, transmittance tails, cdf) are compared absolutely.
# alpha = 1 - exp(-x) cancels to ~6e-8 absolute in empty space whichever exp is used
# (CPU SLEEF vs GPU ocml differ in the last ulp of exp), and sample_pdf divides those
# weights by a sum that can be as small as 62 * 1e-5: cdf noise up to ~1e-3 is inherent
# to the reference's fp32 formula, not to this implementation.
# At the sample_pdf STAGE (identical weights in) cdf and indices are bit-identical to the reference
# (the kernel reproduces ATen's torch.sum order and the fp64 cumsum): those tests use array_equal.
# END TO END the weights themselves differ in the last ulps, so the bounds below are ~3x what the
# GPU runs measure (profiles/r02_pytest_gpu_first.log: fp32 4 flips of 131 072 = 3.1e-5, max |cdf - ref|
# 7.9e-5; bf16x3 4.5e-4; fp16x3 6.1e-5.  Round 1, with an fp64 tree sum in place of ATen's order: 3.0e-4 / 4.7e-4).
W_TOL = 1e-5
CDF_TOL = 3e-4
Z_STD_TOL = 1.5e-3   # a statistic of the sampled depths
FLIP_TOL = 2e-4
FLIP_TOL_X3 = 1.5e-3   # the bf16x3 network output carries 1e-5 instead of 1e-6
FLIP_TOL_SHARP = 1e-3  # sharp scenes (sigma gain 100..300 on every sample, trained weights): measured 1e-4 .. 4.3e-4


def scale_sigma(p, gain=300.0, bias=0.3):
    p = {k: v.clone() for k, v in p.items()}
    p["alpha_linear.weight"] = p["alpha_linear.weight"] * gain
    p["alpha_linear.bias"] = torch.full_like(p["alpha_linear.bias"], bias)
    return p


def device_net(idn, params, dims, dev):
    """(packed weight stream, fold(aud, expr, latent) -> bias block) for oracle-style params."""
    sd = {k: v.to(dev).contiguous() for k, v in params.items()}
    ps = idn.ops.params_struct(sd, dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])
    packed = idn.ops.pack_weights(ps, dev)
    g = lambda t: None if t is None else t.to(dev).contiguous()

    def fold(aud, expr, latent):
        return idn.ops.fold_conditioning(ps, g(aud), g(expr), g(latent), dev)

    fold.keep = sd  # keep device tensors alive as long as the closure lives
    return packed, fold


# --------------------------------------------------------------------------- a5
@pytest.mark.parametrize("name,v", [("c235", dict(dim_aud=64, dim_expr=76, dim_latent=32)),
                                    ("c169", dict(dim_aud=106, dim_expr=0, dim_latent=0)),
                                    ("c127", dict(dim_aud=64, dim_expr=0, dim_latent=0))])
def test_facenerf_fwd_golden(idn, dev, golden, name, v):
    g = golden("facenerf")
    dims = oracle.facenerf_dims(**v)
    params = oracle.xavier_facenerf_params(11, dims)
    packed, fold = device_net(idn, params, dims, dev)
    opt = lambda k: T(g[k]) if k in g else None
    folded = fold(T(g[name + "_aud"]), opt(name + "_expr"), opt(name + "_latent"))
    out = idn.ops.facenerf_fwd(packed, folded, T(g[name + "_x"]).to(dev))
    assert rel_err(out, g[name + "_out"]) < 1e-5


@pytest.mark.parametrize("n", [1, 31, 77, 128, 130, 1000, 40000])
def test_facenerf_fwd_ragged(idn, dev, n):
    """Row counts that do not fill a 32-point wave / 128-point tile / one pass per CU."""
    dims = oracle.facenerf_dims()
    params = scale_sigma(oracle.xavier_facenerf_params(5, dims), 30.0, 0.1)
    rs = np.random.RandomState(n)
    x = T(rs.uniform(-1, 1, size=(n, 90)).astype(np.float32))
    aud, expr, lat = (T(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
    with torch.no_grad():
        ref = oracle.facenerf_forward(params, x, aud, expr, lat, dims)
    packed, fold = device_net(idn, params, dims, dev)
    out = idn.ops.facenerf_fwd(packed, fold(aud, expr, lat), x.to(dev))
    assert out.shape == (n, 4)
    assert rel_err(out, ref) < 1e-5


def test_facenerf_module_dropin(idn, dev, golden):
    """FaceNeRF module: reference state_dict in, reference forward signature, golden out."""
    g = golden("facenerf")
    dims = oracle.facenerf_dims()
    net = idn.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76).to(dev)
    missing = net.load_state_dict(oracle.xavier_facenerf_params(11, dims), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    with torch.no_grad():
        out = net(T(g["c235_x"]).to(dev), T(g["c235_aud"]).to(dev), T(g["c235_expr"]).to(dev),
                  T(g["c235_latent"]).to(dev))
    assert rel_err(out, g["c235_out"]) < 1e-5
    with torch.no_grad(), pytest.raises(RuntimeError):
        net(T(g["c235_x"]).to(dev), T(g["c235_aud"]).to(dev), None, T(g["c235_latent"]).to(dev))
    # a weight update must be picked up (packed stream is rebuilt on version change)
    with torch.no_grad():
        net.rgb_linear.bias.add_(1.0)
        out2 = net(T(g["c235_x"]).to(dev), T(g["c235_aud"]).to(dev), T(g["c235_expr"]).to(dev),
                   T(g["c235_latent"]).to(dev))
    assert rel_err(out2[:, :3] - 1.0, g["c235_out"][:, :3]) < 1e-5


# --------------------------------------------------------------------------- a2 + a4
def test_query_rays_matches_oracle(idn, dev):
    dims = oracle.facenerf_dims()
    params = scale_sigma(oracle.xavier_facenerf_params(7, dims))
    syn = oracle.synthetic_frame(16, 16, seed=3, dims=dims)
    ro, rd = oracle.camera_rays(16, 16, syn["focal"], syn["c2w"])
    rays = oracle.ray_records(ro, rd, NEAR, FAR)
    z = oracle.coarse_depths(rays[:, 6:7], rays[:, 7:8], 64).contiguous()
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    with torch.no_grad():
        ref = oracle.render_oracle._query(params, pts, rays[:, 8:11], syn["aud"], syn["expr"], syn["latent"], dims)
    packed, fold = device_net(idn, params, dims, dev)
    folded = fold(syn["aud"], syn["expr"], syn["latent"])
    raw = idn.ops.query_rays_fwd(packed, folded, rays.to(dev), z.to(dev))
    assert rel_err(raw, ref) < 2e-5
    raw2 = idn.ops.query_points_fwd(packed, folded, pts.contiguous().to(dev), rays[:, 8:11].contiguous().to(dev))
    assert torch.equal(raw, raw2)  # same points, same encodings, same arithmetic


# --------------------------------------------------------------------------- a1, a3
def test_frame_rays_golden(idn, dev, golden):
    g = golden("frame32")
    syn = oracle.synthetic_frame(32, 32, seed=0)
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    assert rel_err(rays, g["rays"]) < 1e-6
    band = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, row0=8, nrows=5, device=dev)
    assert torch.equal(band, rays.reshape(32, 32, 11)[8:13].reshape(-1, 11))


@pytest.mark.parametrize("H,W,rows", [(32, 32, None), (80, 450, None), (200, 200, (20, 200)), (512, 512, (64, 192))])
def test_frame_mode_render_equals_the_ray_record_path(idn, dev, golden, H, W, rows):
    """`idealnerf_render_frame_fwd` takes (c2w, row band, H, W, focal, near, far) instead of a materialised
    rays[n, 11] (helper.py:228-243, audio_exp_nerf.py:396-427 happen on the device, one internal pass at a time).  Every output
    and every tap equals the ray-record call's BIT FOR BIT -- also where an internal 32 768-ray pass ends in the middle of a
    row (W = 450: the reference's frame width) and for a rank's row band -- and the records themselves are the reference's
    (frame32.npz)."""
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(H, W, seed=0, dims=dims)
    cond = (syn["aud"], syn["expr"], syn["latent"])
    r0, r1 = rows or (0, H)
    rays = idn.ops.frame_rays(syn["c2w"], H, W, syn["focal"], NEAR, FAR, row0=r0, nrows=r1 - r0, device=dev)
    bc = syn["bc"][r0:r1].reshape(-1, 3).contiguous().to(dev)
    t, u = torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev)
    fc, ff = fold_c(*cond), fold_f(*cond)
    taps = (H * W <= 40000)     # (the debug taps of the 512 x 512 band would be 0.4 GB)
    a = idn.ops.render_rays_fwd(rays, bc, pk_c, fc, pk_f, ff, t, u, 128, taps=taps)
    frame = idn.ops.make_frame(syn["c2w"], H, W, syn["focal"], NEAR, FAR, r0, r1 - r0)
    b = idn.ops.render_rays_fwd(None, bc, pk_c, fc, pk_f, ff, t, u, 128, taps=taps, frame=frame)
    assert rays.shape[0] > 32768 or H == 32      # (all but the golden frame span more than one internal pass)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    if taps:
        assert torch.equal(b["tap_rays"], rays)
    if H == 32:
        assert rel_err(b["tap_rays"], golden("frame32")["rays"]) < 1e-6
    with pytest.raises(idn._lib.IdealNerfError):
        idn.ops.render_rays_fwd(rays, bc, pk_c, fc, pk_f, ff, t, u, 128, frame=frame)       # both a camera and records
    with pytest.raises(idn._lib.IdealNerfError):
        idn.ops.render_rays_fwd(None, bc[:-1].contiguous(), pk_c, fc, pk_f, ff, t, u, 128, frame=frame)   # background of another band


def test_coarse_depths_bit_exact(idn, dev, golden):
    g = golden("rays64_jitter")
    f = golden("frame32")
    rays = T(f["rays"])[T(g["sel"])].contiguous()
    t = torch.linspace(0.0, 1.0, 64)
    z = idn.ops.coarse_depths(rays.to(dev), t.to(dev), T(g["t_rand"]).to(dev))
    np.testing.assert_array_equal(z.cpu().numpy(), g["z_coarse"])
    z0 = idn.ops.coarse_depths(T(f["rays"]).to(dev), t.to(dev))
    np.testing.assert_array_equal(z0.cpu().numpy(), f["tap_z_coarse"])


# --------------------------------------------------------------------------- a6
@pytest.mark.parametrize("S", [64, 192])
def test_composite_golden(idn, dev, golden, S):
    g = golden("raw2outputs")
    k = lambda n: T(g[f"s{S}_{n}"]).to(dev)
    rays = torch.zeros((64, 11), device=dev)
    rays[:, 3:6] = k("d")
    o = idn.ops.composite_fwd(k("raw"), k("z"), rays, k("bc"), with_fg=True)
    assert rel_err(o["weights"], g[f"s{S}_weights"]) < 2e-6
    assert rel_err(o["rgb_map"], g[f"s{S}_rgb_map"]) < 2e-6
    assert rel_err(o["rgb_fg"], g[f"s{S}_rgb_fg"]) < 2e-6
    assert rel_err(o["depth_map"], g[f"s{S}_depth"]) < 2e-6
    assert rel_err(o["acc_map"], g[f"s{S}_acc"]) < 2e-6
    assert rel_err(o["disp_map"], g[f"s{S}_disp"]) < 2e-6
    np.testing.assert_array_equal(o["last_weight"].cpu().numpy(), o["weights"][:, -1].cpu().numpy())


@pytest.mark.parametrize("S", [2, 3, 63, 65, 100, 129, 255, 256])
def test_composite_ragged_sample_counts(idn, dev, S):
    rs = np.random.RandomState(S)
    n = 37
    raw = T(rs.standard_normal((n, S, 4)).astype(np.float32))
    raw[..., 3] *= 30.0
    z = torch.sort(T(rs.uniform(NEAR, FAR, size=(n, S)).astype(np.float32)), dim=-1)[0]
    d = T(rs.standard_normal((n, 3)).astype(np.float32))
    bc = T(rs.uniform(0, 1, size=(n, 3)).astype(np.float32))
    ref = oracle.composite(raw, z, d, bc, with_fg=True)
    rays = torch.zeros((n, 11))
    rays[:, 3:6] = d
    o = idn.ops.composite_fwd(raw.to(dev), z.to(dev), rays.to(dev), bc.to(dev), with_fg=True)
    for key, r in zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map", "rgb_fg"), ref):
        assert rel_err(o[key], r) < 5e-6, key


# --------------------------------------------------------------------------- a7 (the bit-exact boundary)
@pytest.mark.parametrize("mode", ["det", "rnd"])
def test_invert_cdf_bit_exact(idn, dev, golden, mode):
    g = golden("sample_pdf")
    u = T(g[mode + "_u"])
    u_dev = u[0].contiguous().to(dev) if mode == "det" else u.to(dev)  # det: one shared linspace row
    zs, inds = idn.ops.invert_cdf(T(g[mode + "_cdf"]).to(dev), T(g["bins"]).to(dev), u_dev)
    np.testing.assert_array_equal(inds.cpu().numpy(), g[mode + "_inds"])
    np.testing.assert_array_equal(zs.cpu().numpy(), g[mode + "_samples"])


def _z_from_bins(bins):
    """64 coarse depths whose midpoints are (to rounding) the golden bins: z0 free, z_{k+1} = 2 b_k - z_k."""
    bins = bins.astype(np.float64)
    z = np.empty((bins.shape[0], bins.shape[1] + 1), dtype=np.float64)
    z[:, 0] = bins[:, 0] - 1e-3
    for k in range(bins.shape[1]):
        z[:, k + 1] = 2 * bins[:, k] - z[:, k]
    return z.astype(np.float32)


@pytest.mark.parametrize("mode", ["det", "rnd"])
def test_sample_pdf_golden(idn, dev, golden, mode):
    """The sample_pdf stage on the reference's captured inputs: identical weights in => cdf and
    importance indices bit-identical out (north_star: "bit-exact on importance-sample indices")."""
    g = golden("sample_pdf")
    z32 = _z_from_bins(g["bins"])
    w = np.zeros((64, 64), dtype=np.float32)
    w[:, 1:-1] = g["weights"]
    u = T(g[mode + "_u"])
    u_dev = u[0].contiguous().to(dev) if mode == "det" else u.to(dev)
    o = idn.ops.sample_pdf_fwd(T(z32).to(dev), T(w).to(dev), u_dev, 128)
    np.testing.assert_array_equal(o["cdf"].cpu().numpy(), g[mode + "_cdf"])
    flips = (o["inds"].cpu().numpy() != g[mode + "_inds"]).mean()
    print(f"\nsample_pdf[{mode}]: stage index flip rate vs reference = {flips:.3e} (cdf bit-identical)")
    np.testing.assert_array_equal(o["inds"].cpu().numpy(), g[mode + "_inds"])
    # merged depths are exactly the sorted union of what the kernel itself produced
    ref_sorted = torch.sort(torch.cat([T(z32).to(dev), o["z_samples"]], -1), -1)[0]
    assert torch.equal(o["z_fine"], ref_sorted)
    assert rel_err(o["z_std"], torch.std(o["z_samples"], dim=-1, unbiased=False)) < 1e-5


@pytest.mark.parametrize("mode", ["det", "rnd"])
def test_sample_pdf_bins_entry_bit_exact(idn, dev, golden, mode):
    """helper.sample_pdf's own argument list (bins, interior weights): cdf, indices AND samples bit-identical."""
    g = golden("sample_pdf")
    u = T(g[mode + "_u"])
    u_dev = u[0].contiguous().to(dev) if mode == "det" else u.to(dev)
    o = idn.ops.sample_pdf_bins_fwd(T(g["bins"]).to(dev), T(g["weights"]).to(dev), u_dev)
    np.testing.assert_array_equal(o["cdf"].cpu().numpy(), g[mode + "_cdf"])
    np.testing.assert_array_equal(o["inds"].cpu().numpy(), g[mode + "_inds"])
    np.testing.assert_array_equal(o["z_samples"].cpu().numpy(), g[mode + "_samples"])


def test_sample_pdf_stage_on_the_reference_frame_is_bit_exact(idn, dev, golden):
    """All 1024 rays of the reference's 32x32 frame: its own coarse depths and coarse weights in =>
    its cdf, indices, sampled depths and merged fine depths out, bit for bit."""
    g = golden("frame32")
    u = T(g["tap_u"])[0].contiguous().to(dev)
    o = idn.ops.sample_pdf_fwd(T(g["tap_z_coarse"]).to(dev), T(g["tap_weights_coarse"]).to(dev), u, 128)
    np.testing.assert_array_equal(o["cdf"].cpu().numpy(), g["tap_cdf"])
    flips = (o["inds"].cpu().numpy() != g["tap_inds"].astype(np.int64)).mean()
    print(f"\nframe32 sample_pdf stage (reference weights in): index flip rate = {flips:.3e}")
    np.testing.assert_array_equal(o["inds"].cpu().numpy(), g["tap_inds"].astype(np.int64))
    np.testing.assert_array_equal(o["z_samples"].cpu().numpy(), g["tap_z_samples"])
    np.testing.assert_array_equal(o["z_fine"].cpu().numpy(), g["tap_z_fine"])
    zs = T(g["tap_z_samples"])
    assert rel_err(o["z_std"], torch.std(zs, dim=-1, unbiased=False)) < 1e-5


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 30, 31, 32, 33, 40, 62, 63, 64, 65, 126, 127, 200, 254])
def test_sample_pdf_row_sum_follows_aten_order(idn, dev, K):
    """torch.sum's CPU order for every row length the kernel takes (scalar tail only, one accumulator,
    four accumulators, leftover vectors): the kernel's cdf equals the CPU formula bit for bit."""
    rs = np.random.RandomState(K)
    n, nb = 257, K + 1
    bins = torch.sort(T(rs.uniform(NEAR, FAR, size=(n, nb)).astype(np.float32)), dim=-1)[0]
    w = T((rs.uniform(0, 1, size=(n, K)) ** 8).astype(np.float32))
    u = torch.linspace(0.0, 1.0, 64)
    wp = w + 1e-5
    pdf = wp / torch.sum(wp, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    zs_ref, inds_ref = oracle.invert_cdf(cdf, bins, u.expand(n, 64).contiguous())
    o = idn.ops.sample_pdf_bins_fwd(bins.to(dev), w.to(dev), u.to(dev))
    np.testing.assert_array_equal(o["cdf"].cpu().numpy(), cdf.numpy())
    np.testing.assert_array_equal(o["inds"].cpu().numpy(), inds_ref.numpy())
    np.testing.assert_array_equal(o["z_samples"].cpu().numpy(), zs_ref.numpy())


@pytest.mark.parametrize("S,Ni", [(64, 128), (3, 1), (65, 64), (130, 200), (256, 256)])
def test_march_equals_composite_then_sample_pdf(idn, dev, S, Ni):
    """The fused march kernel (coarse raw2outputs + sample_pdf + merge, weights kept in LDS) against the two
    separate launches it replaces: every output bit for bit, deterministic and random u, with and without the
    weight tap, ragged ray counts."""
    rs = np.random.RandomState(S + Ni)
    n = 1027
    raw = T(rs.standard_normal((n, S, 4)).astype(np.float32))
    raw[..., 3] *= 30.0
    z = torch.sort(T(rs.uniform(NEAR, FAR, size=(n, S)).astype(np.float32)), dim=-1)[0]
    rays = torch.zeros((n, 11))
    rays[:, 3:6] = T(rs.standard_normal((n, 3)).astype(np.float32))
    bc = T(rs.uniform(0, 1, size=(n, 3)).astype(np.float32))
    noise = T((rs.uniform(0, 1, size=(n, S)) * 0.3).astype(np.float32))
    g = lambda t: t.to(dev)
    for u in (torch.linspace(0.0, 1.0, Ni), T(rs.uniform(0, 1, size=(n, Ni)).astype(np.float32))):
        for kw in (dict(), dict(sigma_noise=g(noise), white_bkgd=True)):
            a = idn.ops.composite_fwd(g(raw), g(z), g(rays), g(bc), with_fg=True, with_weights=True, **kw)
            b = idn.ops.sample_pdf_fwd(g(z), a["weights"], g(u), Ni)
            for with_w in (True, False):
                m = idn.ops.march_fwd(g(raw), g(z), g(rays), g(bc), g(u), Ni, with_fg=True, with_weights=with_w, **kw)
                assert ("weights" in m) == with_w
                for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "last_weight", "rgb_fg") + (("weights",) if with_w else ()):
                    assert torch.equal(m[k], a[k]), k
                for k in ("z_samples", "inds", "cdf", "z_fine", "z_std"):
                    assert torch.equal(m[k], b[k]), k


def test_merge_handles_ties_unsorted_samples_and_nan(idn, dev):
    """z_fine == torch.sort(cat[z, z_samples]) whatever branch merges: sorted halves with exact ties
    (the binary-search merge), random u (unsorted samples: rank counting), and NaN depths (torch.sort
    puts them last; every output slot is written)."""
    rs = np.random.RandomState(4)
    n = 300
    z = torch.sort(T(rs.uniform(NEAR, FAR, size=(n, 64)).astype(np.float32)), dim=-1)[0]
    z[::3, 10:20] = z[::3, 10:11]          # runs of equal coarse depths => zero-width bins, equal samples
    w = T((rs.uniform(0, 1, size=(n, 64)) ** 6).astype(np.float32))
    w[::5, 20:40] = 0.0
    for u in (torch.linspace(0.0, 1.0, 128), T(rs.uniform(0, 1, size=(n, 128)).astype(np.float32))):
        o = idn.ops.sample_pdf_fwd(z.to(dev), w.to(dev), u.to(dev), 128)
        ref = torch.sort(torch.cat([z.to(dev), o["z_samples"]], -1), -1)[0]
        assert torch.equal(o["z_fine"], ref)
    zn = z.clone()
    zn[7, 63] = float("nan")
    zn[9, 5] = float("nan")               # a NaN in the middle: that half is no longer ordered
    o = idn.ops.sample_pdf_fwd(zn.to(dev), w.to(dev), torch.linspace(0.0, 1.0, 128).to(dev), 128)
    filled = torch.full_like(o["z_fine"], -1.0)
    ref = torch.sort(torch.cat([zn.to(dev), o["z_samples"]], -1), -1)[0]
    assert torch.equal(torch.isnan(o["z_fine"]), torch.isnan(ref))
    ok = ~torch.isnan(ref)
    assert torch.equal(o["z_fine"][ok], ref[ok]) and filled.shape == ref.shape


def test_sample_pdf_own_cdf_is_bit_exact_boundary(idn, dev):
    """Feed the kernel's own cdf back through the CPU searchsorted: identical indices."""
    rs = np.random.RandomState(9)
    n = 512
    z = torch.sort(T(rs.uniform(NEAR, FAR, size=(n, 64)).astype(np.float32)), dim=-1)[0]
    w = T((rs.uniform(0, 1, size=(n, 64)) ** 6).astype(np.float32))
    u = torch.linspace(0.0, 1.0, 128)
    o = idn.ops.sample_pdf_fwd(z.to(dev), w.to(dev), u.to(dev), 128)
    cdf = o["cdf"].cpu()
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    zs_ref, inds_ref = oracle.invert_cdf(cdf, bins, u.expand(n, 128).contiguous())
    np.testing.assert_array_equal(o["inds"].cpu().numpy(), inds_ref.numpy())
    np.testing.assert_array_equal(o["z_samples"].cpu().numpy(), zs_ref.numpy())


# --------------------------------------------------------------------------- a3-a9 end to end
def _nets(idn, dev):
    dims = oracle.facenerf_dims()
    pc = scale_sigma(oracle.xavier_facenerf_params(2, dims))
    pf = scale_sigma(oracle.xavier_facenerf_params(3, dims))
    return dims, pc, pf, device_net(idn, pc, dims, dev), device_net(idn, pf, dims, dev)


def test_render_frame32_golden(idn, dev, golden):
    g = golden("frame32")
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    cond = (syn["aud"], syn["expr"], syn["latent"])
    t = torch.linspace(0.0, 1.0, 64).to(dev)
    u = torch.linspace(0.0, 1.0, 128).to(dev)
    out = idn.ops.render_rays_fwd(rays, syn["bc"].reshape(-1, 3).to(dev), pk_c, fold_c(*cond), pk_f, fold_f(*cond),
                                  t, u, 128, taps=True)
    np.testing.assert_array_equal(out["tap_z_coarse"].cpu().numpy(), g["tap_z_coarse"])
    assert rel_err(out["tap_raw_coarse"][:128], g["tap_raw_coarse"]) < 2e-5
    assert abs_err(out["tap_weights_coarse"], g["tap_weights_coarse"]) < W_TOL
    assert abs_err(out["tap_cdf"], g["tap_cdf"]) < CDF_TOL
    flips = (out["tap_inds"].cpu().numpy() != g["tap_inds"].astype(np.int64)).mean()
    print(f"\nframe32: end-to-end importance-index flip rate vs reference = {flips:.3e}, "
          f"max |cdf - ref| = {abs_err(out['tap_cdf'], g['tap_cdf']):.2e}")
    assert flips < FLIP_TOL, f"end-to-end importance-index flip rate {flips}"
    # a flipped index moves a sample continuously (the inverse CDF is piecewise linear)
    assert abs_err(out["tap_z_fine"], g["tap_z_fine"]) < 2e-3 * (FAR - NEAR)
    for k, gk in (("rgb_map", "rgb"), ("rgb0", "rgb0")):
        assert rel_err(out[k], g[gk].reshape(-1, 3)) < RGB_TOL, k
    for k, gk in (("disp_map", "disp"), ("acc_map", "acc"), ("disp0", "disp0"), ("acc0", "acc0")):
        assert rel_err(out[k], g[gk].reshape(-1)) < RGB_TOL, k
    assert rel_err(out["z_std"], g["z_std"].reshape(-1)) < Z_STD_TOL
    assert abs_err(out["last_weight"], g["last_weight"].reshape(-1)) < W_TOL
    mse = float(((out["rgb_map"].cpu().numpy() - g["rgb"].reshape(-1, 3)) ** 2).mean())
    assert mse < 1e-10  # PSNR vs reference output > 100 dB


def test_render_rays_jitter_golden(idn, dev, golden):
    g = golden("rays64_jitter")
    f = golden("frame32")
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    sel = T(g["sel"])
    cond = (syn["aud"], syn["expr"], syn["latent"])
    out = idn.ops.render_rays_fwd(T(f["rays"])[sel].contiguous().to(dev),
                                  syn["bc"].reshape(-1, 3)[sel].contiguous().to(dev), pk_c, fold_c(*cond), pk_f,
                                  fold_f(*cond), torch.linspace(0.0, 1.0, 64).to(dev), T(g["u"]).to(dev), 128,
                                  t_rand=T(g["t_rand"]).to(dev), taps=True)
    np.testing.assert_array_equal(out["tap_z_coarse"].cpu().numpy(), g["z_coarse"])
    flips = (out["tap_inds"].cpu().numpy() != g["inds"]).mean()
    print(f"\nrays64_jitter: importance-index flip rate vs reference = {flips:.3e}")
    assert flips < FLIP_TOL
    for k in ("rgb_map", "rgb0", "disp_map", "acc_map"):
        assert rel_err(out[k], g[k]) < RGB_TOL, k
    assert rel_err(out["z_std"], g["z_std"]) < Z_STD_TOL
    assert abs_err(out["last_weight"], g["last_weight"]) < W_TOL
    # random u: the merge is a real sort; the result must be sorted and a permutation of the inputs
    zf = out["tap_z_fine"]
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())
    assert torch.equal(zf, torch.sort(torch.cat([out["tap_z_coarse"], out["tap_z_samples"]], -1), -1)[0])


def test_render_rays_coarse_only_golden(idn, dev, golden):
    """BASELINE config 1: 256 rays, N_importance = 0."""
    g = golden("rays256_coarse_only")
    f = golden("frame32")
    dims, pc, pf, (pk_c, fold_c), _ = _nets(idn, dev)
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    sel = T(g["sel"])
    cond = (syn["aud"], syn["expr"], syn["latent"])
    out = idn.ops.render_rays_fwd(T(f["rays"])[sel].contiguous().to(dev),
                                  syn["bc"].reshape(-1, 3)[sel].contiguous().to(dev), pk_c, fold_c(*cond), None, None,
                                  torch.linspace(0.0, 1.0, 64).to(dev), None, 0)
    assert set(out) == {"rgb_map", "disp_map", "acc_map"}
    for k in out:
        assert rel_err(out[k], g[k]) < RGB_TOL, k


def test_network_eval_forward_golden(idn, dev, golden):
    """The reference's call: network([data, global_step, dataset_size]) in eval mode
    (eval_aud_exp_nerf.py:489) -> [rgb, disp, acc, last_weight, extras]."""
    from idealnerf_amd.audio_exp_nerf import Network
    from idealnerf_amd.helper import RenderConfig
    g = golden("frame32")
    dims = oracle.facenerf_dims()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    cfg = RenderConfig(perturb=0.0, chunk=512, near=NEAR, far=FAR)
    net = Network(32, 32, syn["focal"], NEAR, FAR, 512, None, 64, 128, args=cfg).to(dev)
    net.face_nerf_coarse.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(2, dims)))
    net.face_nerf_fine.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(3, dims)))
    net.eval()
    with torch.no_grad():
        rgb, disp, acc, last_w, extras = net.render_dynamic_face(
            32, 32, syn["focal"], expr=syn["expr"].to(dev), poses=syn["c2w"], latent_code=syn["latent"].to(dev),
            render_poses=syn["c2w"][:3, :4], chunk=512, near=NEAR, far=FAR, rays=None, bc_rgb=syn["bc"].to(dev),
            aud_para=syn["aud"].to(dev))
    assert rgb.shape == (32, 32, 3) and set(extras) == {"rgb0", "disp0", "acc0", "z_std"}
    assert rel_err(rgb, g["rgb"]) < RGB_TOL
    assert rel_err(extras["rgb0"], g["rgb0"]) < RGB_TOL
    assert abs_err(last_w, g["last_weight"]) < W_TOL
    # N_importance = 0 through render_dynamic_face raises KeyError('last_weight') upstream too (:437)
    cfg0 = RenderConfig(perturb=0.0, chunk=512, N_importance=0)
    net0 = Network(32, 32, syn["focal"], NEAR, FAR, 512, None, 64, 0, args=cfg0).to(dev)
    with torch.no_grad(), pytest.raises(KeyError):
        net0.render_dynamic_face(32, 32, syn["focal"], expr=syn["expr"].to(dev), poses=syn["c2w"],
                                 latent_code=syn["latent"].to(dev), render_poses=syn["c2w"][:3, :4], chunk=512,
                                 near=NEAR, far=FAR, bc_rgb=syn["bc"].to(dev), aud_para=syn["aud"].to(dev))


# --------------------------------------------------------------------------- full-size properties
def test_full_size_band_properties(idn, dev):
    """BASELINE config 2 sizes (512x512, 64+128) on a 64-row band (one rank's share at 8 GPUs):
    size-independent properties -- weights sum to one, depths sorted, the result does not
    depend on how the rays are batched, and two runs agree bit for bit."""
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(512, 512, seed=0, dims=dims)
    cond = (syn["aud"], syn["expr"], syn["latent"])
    fc, ff = fold_c(*cond), fold_f(*cond)
    rays = idn.ops.frame_rays(syn["c2w"], 512, 512, syn["focal"], NEAR, FAR, row0=192, nrows=80, device=dev)
    bc = syn["bc"][192:272].reshape(-1, 3).contiguous().to(dev)
    t = torch.linspace(0.0, 1.0, 64).to(dev)
    u = torch.linspace(0.0, 1.0, 128).to(dev)
    run = lambda r, b, **kw: idn.ops.render_rays_fwd(r, b, pk_c, fc, pk_f, ff, t, u, 128, **kw)
    full = run(rays, bc, taps=True)  # 40960 rays: crosses the library's internal 32768-ray pass boundary
    assert torch.isfinite(full["rgb_map"]).all()
    assert float((full["acc_map"] - 1.0).abs().max()) < 1e-5
    assert float((full["tap_weights_fine"].sum(-1) - 1.0).abs().max()) < 1e-5
    zf = full["tap_z_fine"]
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())
    assert float(zf.min()) >= NEAR - 1e-6 and float(zf.max()) <= FAR + 1e-6
    inds = full["tap_inds"]
    assert int(inds.min()) >= 1 and int(inds.max()) <= 63
    assert float((full["rgb_map"] - bc).abs().mean()) > 0.02  # the volume is visible
    again = run(rays, bc)
    assert torch.equal(again["rgb_map"], full["rgb_map"])
    part = run(rays[1000:1777].contiguous(), bc[1000:1777].contiguous())
    assert torch.equal(part["rgb_map"], full["rgb_map"][1000:1777])
    assert torch.equal(part["z_std"], full["z_std"][1000:1777])
    # CPU oracle on every 80th ray of the band (512 rays; the reference-generated tile of this frame is
    # test_frame512_tile_golden): fixed 1e-4, flips attributed (parity_proof)
    idx = torch.arange(0, rays.shape[0], 80)
    with torch.no_grad():
        ref = oracle.render_rays(rays[idx].cpu(), bc[idx].cpu(), pc, pf, *cond, dims=dims, taps=True)
    sub = {k: v[idx.to(dev)] for k, v in full.items()}
    r_sub, bc_sub = rays[idx.to(dev)].contiguous(), bc[idx.to(dev)].contiguous()
    prove_render(idn, "full-size band", sub, ref, pk_f, ff, r_sub, bc_sub,
                 lambda z: oracle_fine_pass(pf, dims, r_sub, bc_sub, *cond, z), FLIP_TOL)
    assert rel_err(sub["rgb0"], ref["rgb0"]) < RGB_TOL


# --------------------------------------------------------------------------- a12: training step
def grad_err(got, ref, scale_ref=None):
    """max |got - ref| relative to the largest reference entry of the layer (for a bias: of its
    weight gradient too, so a single-element gradient that cancels to ~0 is not divided by ~0)."""
    g = got.detach().cpu().double()
    r = ref.detach().cpu().double()
    scale = float(r.abs().max())
    if scale_ref is not None:
        scale = max(scale, float(scale_ref.detach().cpu().abs().max()))
    return float((g - r).abs().max()) / max(scale, 1e-30)


def l2_err(got, ref):
    g = got.detach().cpu().double()
    r = ref.detach().cpu().double()
    return float((g - r).norm() / r.norm().clamp_min(1e-30))


# End-to-end gradients against CPU autograd are compared by relative L2 plus a loose max bound:
# a hidden unit whose pre-activation is within rounding of zero gets a different ReLU mask on
# MFMA than on the CPU BLAS (and a flipped importance index moves a fine sample), which moves
# single rows of single tensors at the 1e-3..1e-2 level when only a few dozen rays carry the
# gradient.  The arithmetic itself is pinned by test_backward_kernels_vs_fp64_on_saved_activations
# (same activations, same masks: 1e-6).
GRAD_L2_COARSE, GRAD_L2_FINE, GRAD_MAX = 2e-3, 1e-2, 5e-2


def check_grads(named_params, ref, fine, ctx=""):
    for name, prm in named_params:
        if name.startswith("feature_linear"):
            continue
        r = ref[name].grad
        w_ref = ref[name.replace(".bias", ".weight")].grad
        assert l2_err(prm.grad, r) < (GRAD_L2_FINE if fine else GRAD_L2_COARSE) or grad_err(prm.grad, r, w_ref) < 1e-5, (ctx, name)
        assert grad_err(prm.grad, r, w_ref) < GRAD_MAX, (ctx, name)


def _train_net(idn, dev, n_importance=128):
    torch.manual_seed(1234)  # the audio nets are torch-initialised: keep every process on the same instance
    from idealnerf_amd.audio_exp_nerf import Network
    from idealnerf_amd.helper import RenderConfig
    dims = oracle.facenerf_dims()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    cfg = RenderConfig(perturb=0.0, chunk=512, near=NEAR, far=FAR, N_importance=n_importance)
    net = Network(32, 32, syn["focal"], NEAR, FAR, 512, None, 64, n_importance, args=cfg).to(dev)
    net.face_nerf_coarse.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(2, dims)))
    net.face_nerf_fine.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(3, dims)))
    net.train()
    return net, syn


def test_train_step_gradients_golden(idn, dev, golden):
    """One training step's loss and gradients (audio_exp_nerf.py:534-550) against autograd
    through the reference itself."""
    from idealnerf_amd.helper import img2mse
    g = golden("train_step")
    f = golden("frame32")
    net, syn = _train_net(idn, dev)
    sel = T(g["sel"])
    rays = T(f["rays"])[sel].contiguous().to(dev)
    bc = syn["bc"].reshape(-1, 3)[sel].contiguous().to(dev)
    aud = syn["aud"].to(dev).requires_grad_(True)
    lat = syn["latent"].to(dev).requires_grad_(True)
    tgt = T(g["target"]).to(dev)
    ret = net.render_rays(rays, bc, aud, syn["c2w"], lat, syn["expr"].to(dev))
    assert rel_err(ret["rgb_map"], g["rgb_map"]) < RGB_TOL and rel_err(ret["rgb0"], g["rgb0"]) < RGB_TOL
    img_loss = img2mse(ret["rgb_map"], tgt)
    loss = img_loss + img2mse(ret["rgb0"], tgt) + 10 * (torch.norm(lat) * 0.0005)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    GRAD_TOL = 2e-3  # relative to the largest entry of each gradient tensor
    errs = {"aud": rel_err(aud.grad, g["g_aud"]), "latent": rel_err(lat.grad, g["g_latent"])}
    for tag, m in (("c", net.face_nerf_coarse), ("f", net.face_nerf_fine)):
        named = dict(m.named_parameters())
        for k in ("pts_linears.0.weight", "pts_linears.0.bias", "pts_linears.5.weight", "pts_linears.7.weight",
                  "views_linears.0.weight", "views_linears.2.bias", "alpha_linear.weight", "alpha_linear.bias",
                  "rgb_linear.weight", "rgb_linear.bias"):
            errs[f"{tag}.{k}"] = rel_err(named[k].grad, g[f"g_{tag}_{k}"])
        assert named["feature_linear.weight"].grad is None
    print("\ntrain-step gradient errors (max abs / max |ref|):", {k: f"{v:.1e}" for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if not v < GRAD_TOL}
    assert not bad, bad


def test_train_step_matches_oracle_autograd_ragged(idn, dev):
    """Ray count that fills neither a wave nor a 128-point tile; N_importance = 0 variant too."""
    from idealnerf_amd.helper import img2mse
    for n_rays, ni in ((37, 128), (50, 0)):
        net, syn = _train_net(idn, dev, ni)
        rs = np.random.RandomState(n_rays)
        ro, rd = oracle.camera_rays(32, 32, syn["focal"], syn["c2w"])
        rays_all = oracle.ray_records(ro, rd, NEAR, FAR)
        sel = T(rs.choice(1024, size=n_rays, replace=False))
        rays, bc = rays_all[sel].contiguous(), syn["bc"].reshape(-1, 3)[sel].contiguous()
        tgt = T(rs.uniform(0, 1, size=(n_rays, 3)).astype(np.float32))
        # oracle autograd
        dims = oracle.facenerf_dims()
        pc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.face_nerf_coarse.state_dict().items()}
        pf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.face_nerf_fine.state_dict().items()}
        aud_o = syn["aud"].clone().requires_grad_(True)
        lat_o = syn["latent"].clone().requires_grad_(True)
        out_o = oracle.render_rays(rays, bc, pc, pf if ni else None, aud_o, syn["expr"], lat_o, n_importance=ni, dims=dims)
        loss_o, _ = oracle.train_loss(out_o, tgt, lat_o)
        loss_o.backward()
        # HIP
        aud = syn["aud"].to(dev).requires_grad_(True)
        lat = syn["latent"].to(dev).requires_grad_(True)
        ret = net.render_rays(rays.to(dev), bc.to(dev), aud, syn["c2w"], lat, syn["expr"].to(dev))
        loss = img2mse(ret["rgb_map"], tgt.to(dev))
        if ni:
            loss = loss + img2mse(ret["rgb0"], tgt.to(dev))
        loss = loss + 10 * (torch.norm(lat) * 0.0005)
        loss.backward()
        assert abs(float(loss) - float(loss_o)) < 1e-5 * abs(float(loss_o))
        assert l2_err(aud.grad, aud_o.grad) < GRAD_L2_FINE and l2_err(lat.grad, lat_o.grad) < GRAD_L2_FINE
        check_grads(net.face_nerf_coarse.named_parameters(), pc, False, n_rays)
        if ni:
            check_grads(net.face_nerf_fine.named_parameters(), pf, True, n_rays)


def test_relu_backward_at_exactly_zero_matches_torch(idn, dev):
    """A unit whose pre-activation is exactly +0.0 is OFF in torch's relu backward (grad * (result > 0)).  pts_linears.3
    with zero weights and zero bias puts all 256 of its pre-activations at +0.0 for every point: no gradient may pass
    -- d(pts_linears.3.bias) = 0 and everything upstream of it gets zero gradient -- exactly as the reference's autograd
    has it (the ReLU masks record "pre-activation <= 0", not the sign bit)."""
    from idealnerf_amd.helper import img2mse
    net, syn = _train_net(idn, dev)
    with torch.no_grad():
        for m in (net.face_nerf_coarse, net.face_nerf_fine):
            m.pts_linears[3].weight.zero_()
            m.pts_linears[3].bias.zero_()
    rs = np.random.RandomState(3)
    ro, rd = oracle.camera_rays(32, 32, syn["focal"], syn["c2w"])
    sel = T(rs.choice(1024, size=64, replace=False))
    rays, bc = oracle.ray_records(ro, rd, NEAR, FAR)[sel].contiguous(), syn["bc"].reshape(-1, 3)[sel].contiguous()
    tgt = T(rs.uniform(0, 1, size=(64, 3)).astype(np.float32))
    dims = oracle.facenerf_dims()
    pc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.face_nerf_coarse.state_dict().items()}
    pf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.face_nerf_fine.state_dict().items()}
    aud_o, lat_o = syn["aud"].clone().requires_grad_(True), syn["latent"].clone().requires_grad_(True)
    loss_o, _ = oracle.train_loss(oracle.render_rays(rays, bc, pc, pf, aud_o, syn["expr"], lat_o, dims=dims), tgt, lat_o)
    loss_o.backward()
    assert float(pc["pts_linears.3.bias"].grad.abs().max()) == 0.0 and float(pc["pts_linears.1.weight"].grad.abs().max()) == 0.0   # torch: off
    aud, lat = syn["aud"].to(dev).requires_grad_(True), syn["latent"].to(dev).requires_grad_(True)
    ret = net.render_rays(rays.to(dev), bc.to(dev), aud, syn["c2w"], lat, syn["expr"].to(dev))
    loss = img2mse(ret["rgb_map"], tgt.to(dev)) + img2mse(ret["rgb0"], tgt.to(dev)) + 10 * (torch.norm(lat) * 0.0005)
    loss.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-5 * abs(float(loss_o))
    for m, ref in ((net.face_nerf_coarse, pc), (net.face_nerf_fine, pf)):
        named = dict(m.named_parameters())
        for k in ("pts_linears.3.bias", "pts_linears.3.weight", "pts_linears.2.weight", "pts_linears.0.weight"):
            assert float(named[k].grad.abs().max()) == 0.0, k          # nothing passes a unit that sits at +0.0
        assert float(named["pts_linears.4.weight"].grad.abs().max()) == 0.0      # its input is relu(0) = 0 for every point
        for k in ("pts_linears.4.bias", "pts_linears.5.weight", "pts_linears.7.weight", "rgb_linear.weight"):
            assert l2_err(named[k].grad, ref[k].grad) < GRAD_L2_FINE, k
    # the conditioning still reaches the loss through pts_linears.5's re-injected input
    assert l2_err(aud.grad, aud_o.grad) < GRAD_L2_FINE


def test_training_on_the_fp32_pipe_still_passes_the_gradient_tests(dev):
    """The fallback arm: IDN_TRAIN_PRECISION=f32 (forward on the fp32 MFMA kernel) + IDN_BACKWARD_PIPE=f32 (fp32 delta
    chain and fp32 256 x 256 GEMMs) are read once per process, so the gradient tests run again in a fresh one."""
    import subprocess
    import sys
    import signal
    env = dict(os.environ, IDN_TRAIN_PRECISION="f32", IDN_BACKWARD_PIPE="f32")
    # its own process group and a limit below the suite's: a child that hangs is killed by group id, it does not keep the GPU
    child = subprocess.Popen([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                              "train_step_gradients_golden or backward_kernels_vs_fp64 or train_step_matches_oracle_autograd_ragged or exactly_zero"],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        out, err = child.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, signal.SIGKILL)
        child.wait()
        pytest.fail("the fp32-pipe child run did not finish within 240 s")
    assert child.returncode == 0 and "4 passed" in out, out[-3000:] + err[-2000:]


def test_train_step_is_bit_reproducible_at_bench_scale(idn, dev):
    """The training kernels keep row stores in flight across slice barriers behind hand-counted `vmcnt` waits (weight
    pieces are waited for, the stores issued after them are not) and reload row registers in place behind counted waits:
    a count that reached back into a piece would read a weight slice before it has landed -- sometimes.  Nothing in the
    step uses atomics, so every run of the same step must give the same bits: 12 runs of forward + backward on 3072 rays
    x (64 + 192) points (the bench's step: every CU busy, 36 GB through HBM) against the first."""
    from idealnerf_amd.helper import img2mse
    net, syn = _train_net(idn, dev)
    rs = np.random.RandomState(7)
    H = W = 96
    rec = idn.ops.frame_rays(syn["c2w"], H, W, syn["focal"] * 3, NEAR, FAR, device=dev)
    sel = T(rs.choice(H * W, size=3072, replace=False)).to(dev)
    rays = rec[sel].contiguous()
    bc = T(rs.uniform(0, 1, size=(3072, 3)).astype(np.float32)).to(dev)
    tgt = T(rs.uniform(0, 1, size=(3072, 3)).astype(np.float32)).to(dev)
    first = None
    for run in range(12):
        aud = syn["aud"].to(dev).requires_grad_(True)
        lat = syn["latent"].to(dev).requires_grad_(True)
        for p_ in net.parameters():
            p_.grad = None
        ret = net.render_rays(rays, bc, aud, syn["c2w"], lat, syn["expr"].to(dev))
        loss = img2mse(ret["rgb_map"], tgt) + img2mse(ret["rgb0"], tgt) + 10 * (torch.norm(lat) * 0.0005)
        loss.backward()
        got = {"loss": loss.detach().clone(), "rgb": ret["rgb_map"].detach().clone(), "aud": aud.grad.clone(), "lat": lat.grad.clone()}
        for tag, m in (("c", net.face_nerf_coarse), ("f", net.face_nerf_fine)):
            for k, p_ in m.named_parameters():
                if p_.grad is not None:
                    got[f"{tag}.{k}"] = p_.grad.clone()
        assert all(bool(torch.isfinite(v).all()) for v in got.values())
        if first is None:
            first = got
        else:
            diff = [k for k in first if not torch.equal(first[k], got[k])]
            assert not diff, (run, diff)


# --------------------------------------------------------------------------- a11: head + torso composite
def _torso_setup(idn, dev, n=48):
    torch.manual_seed(4321)  # the audio net is torch-initialised: same instance in every process
    from idealnerf_amd.train_torso import Network
    from idealnerf_amd.helper import RenderConfig
    rs = np.random.RandomState(5)
    syn = oracle.synthetic_frame(32, 32, seed=4)
    cfg = RenderConfig(perturb=0.0, chunk=512, near=NEAR, far=FAR, dim_expr=79)
    net = Network(32, 32, syn["focal"], NEAR, FAR, 512, 64, 128, args=cfg).to(dev)
    dh = oracle.facenerf_dims(dim_aud=64, dim_expr=79, dim_latent=32)
    dt = oracle.facenerf_dims(dim_aud=106, dim_expr=0, dim_latent=0)
    P = dict(hc=scale_sigma(oracle.xavier_facenerf_params(21, dh), 100.0, 0.2),
             hf=scale_sigma(oracle.xavier_facenerf_params(22, dh), 100.0, 0.2),
             # semi-transparent torso: the head reaches the pixel through last_weight_torso, which must be
             # O(0.1..1) for the head gradients to be meaningful (an opaque torso leaves ~1e-9 factors whose
             # fp32 relative accuracy is ~1e-2 on any implementation)
             tc=scale_sigma(oracle.xavier_facenerf_params(23, dt), 4.0, -0.2),
             tf=scale_sigma(oracle.xavier_facenerf_params(24, dt), 4.0, -0.2))
    net.face_nerf_coarse.load_state_dict(P["hc"]); net.face_nerf_fine.load_state_dict(P["hf"])
    net.torso_coarse_nerf.load_state_dict(P["tc"]); net.torso_fine_nerf.load_state_dict(P["tf"])
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
    pose0 = torch.eye(4); pose0[:3, 3] = torch.tensor([0.02, -0.01, 0.9])
    ro, rd = oracle.camera_rays(32, 32, syn["focal"], pose[:3, :4])
    ro0, rd0 = oracle.camera_rays(32, 32, syn["focal"], pose0[:3, :4])
    sel = T(rs.choice(1024, size=n, replace=False))
    pick = lambda a: a.reshape(-1, 3)[sel]
    data = dict(batch_rays=torch.stack([pick(ro), pick(rd)], 0), batch_rays_torso=torch.stack([pick(ro0), pick(rd0)], 0),
                bg=pick(syn["bc"]), auds=T(rs.standard_normal((4, 16, 29)).astype(np.float32)), pose=pose,
                expr=T(rs.standard_normal(79).astype(np.float32)), latent=torch.ones(32), target=T(rs.uniform(0, 1, (n, 3)).astype(np.float32)))
    return net, syn, P, (dh, dt), data


def _torso_oracle(net, P, dims, data, grad=False, parts=False):
    dh, dt = dims
    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    aud_net = type(net.aud_net)(64, 16); aud_net.load_state_dict(cpu(net.aud_net))
    with torch.set_grad_enabled(grad):
        aud_feature = aud_net(data["auds"][1:2])
        aud_torso = oracle.torso_signal(aud_feature, data["pose"])
        rec = lambda r: oracle.ray_records(r[0], r[1], NEAR, FAR)
        head = oracle.render_rays(rec(data["batch_rays"]), data["bg"], P["hc"], P["hf"], aud_feature, data["expr"],
                                  data["latent"], dims=dh, with_fg=True, taps=parts)
        torso = oracle.render_rays(rec(data["batch_rays_torso"]), data["bg"], P["tc"], P["tf"], aud_torso, None, None,
                                   dims=dt, with_fg=True, taps=parts)
        if parts:
            return oracle.head_torso_composite(head, torso), head, torso
        return oracle.head_torso_composite(head, torso), aud_net


def _torso_hip_parts(net, d, dev):
    """The two renders behind train_torso.Network.forward (ray-batch branch), with the debug taps: what forward
    composites, by the same calls (`_render` is what `render_pair` -> `_batchify` reaches)."""
    g = lambda t: t.to(device=dev, dtype=torch.float32)
    with torch.no_grad():
        aud_feature = net.aud_net(g(d["auds"])[1:2])
        aud_torso = net.torso_signal(aud_feature, g(d["pose"]))
        def rec(r):   # the records as render_pair builds them, on the device (same ops => same bits as forward)
            ro, rd = g(r[0]).reshape(-1, 3), g(r[1]).reshape(-1, 3)
            one = torch.ones_like(rd[..., :1])
            return torch.cat([ro, rd, NEAR * one, FAR * one, rd / torch.norm(rd, dim=-1, keepdim=True)], -1).contiguous()
        rays_h, rays_t, bg = rec(d["batch_rays"]), rec(d["batch_rays_torso"]), g(d["bg"]).contiguous()
        expr, lat = g(d["expr"]), g(d["latent"])
        head = net._render(rays_h, bg, aud_feature, lat, expr, net.face_nerf_coarse, net.face_nerf_fine, True, taps=True)
        torso = net._render(rays_t, bg, aud_torso, None, None, net.torso_coarse_nerf, net.torso_fine_nerf, True, taps=True)
        folds = (net.face_nerf_fine.folded_bias(aud_feature, expr, lat), net.torso_fine_nerf.folded_bias(aud_torso, None, None))
    return head, torso, rays_h, rays_t, bg, folds


def prove_head_torso(idn, what, net, P, dims, d, dev, rgb_com, flip_bound):
    """parity_proof's (1)-(3) for the composite rgb_head * last_weight_torso + rgb_fg_torso (train_torso.py:269-270): the
    stage checks on both renders; the composite of the two HIP fine passes against the composite of the two oracle fine
    passes on the same positions (the oracle's, then the HIP path's), fixed 1e-4, every ray."""
    (ref, ref0), ref_h, ref_t = _torso_oracle(net, P, dims, d, parts=True)
    head, torso, rays_h, rays_t, bg, (fold_h, fold_t) = _torso_hip_parts(net, d, dev)
    hip_com = head["rgb_map"] * torso["last_weight"][..., None] + torso["rgb_map_fg"]
    assert torch.equal(hip_com.reshape(rgb_com.shape), torch.as_tensor(rgb_com, device=dev, dtype=torch.float32)), "forward composites something else"
    check_stage(what + " (head)", head, ref_h, 128)
    check_stage(what + " (torso)", torso, ref_t, 128)
    fh, ft = net.face_nerf_fine, net.torso_fine_nerf
    compose = lambda h, t: h["rgb_map"] * t["last_weight"][..., None] + t["rgb_fg"]
    # HIP fine passes on the oracle's positions (vs the oracle's composite) ...
    on_ref = compose(hip_fine_pass(idn, fh.packed_weights(), fold_h, rays_h, bg, ref_h["tap_z_fine"], fh.prec_code),
                     hip_fine_pass(idn, ft.packed_weights(), fold_t, rays_t, bg, ref_t["tap_z_fine"], ft.prec_code, with_fg=True))
    # ... and the oracle's fine passes on the HIP positions (vs the HIP composite)
    with torch.no_grad():
        cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
        aud_net = type(net.aud_net)(64, 16); aud_net.load_state_dict(cpu(net.aud_net))
        aud_feature = aud_net(d["auds"][1:2])
        aud_torso = oracle.torso_signal(aud_feature, d["pose"])
    ora_on_hip = compose(oracle_fine_pass(P["hf"], dims[0], rays_h, bg, aud_feature, d["expr"], d["latent"], head["tap_z_fine"]),
                         oracle_fine_pass(P["tf"], dims[1], rays_t, bg, aud_torso, None, None, torso["tap_z_fine"], with_fg=True))
    (fl_h, rate_h), (fl_t, rate_t) = flipped_rows(head["tap_inds"], ref_h["tap_inds"]), flipped_rows(torso["tap_inds"], ref_t["tap_inds"])
    same = ((head["tap_z_fine"].cpu() == ref_h["tap_z_fine"]).all(1) & (torso["tap_z_fine"].cpu() == ref_t["tap_z_fine"]).all(1)).numpy()
    res = prove(what, (on_ref, ref), (hip_com, ora_on_hip), hip_com, ref, fl_h | fl_t, max(rate_h, rate_t, key=float),
                small_sample_bound(flip_bound, head["tap_inds"].numel()), same)
    return res, ref, ref0


def test_head_torso_composite_matches_oracle(idn, dev):
    net, syn, P, dims, d = _torso_setup(idn, dev)
    net.train()
    x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
         d["expr"][None], d["latent"], torch.tensor([1]))
    with torch.no_grad():
        rgb_com, rgb_com0 = net([x, 0, 4])
    assert rgb_com.shape == (48, 3)
    _, ref, ref0 = prove_head_torso(idn, "head+torso composite", net, P, dims, d, dev, rgb_com, FLIP_TOL_SHARP)
    assert rel_err(rgb_com0, ref0) < RGB_TOL


def test_head_torso_golden(idn, dev, golden):
    """a11 / BASELINE configs[4] against the REFERENCE's own TorsoNeRF code (tests/golden/head_torso.npz: train_torso.py
    Network.forward with run_nerf.raw2outputs' rgb_map_fg and run_nerf_helpers.sample_pdf, 512 rays of the sharp scene):
    the product's forward with the reference's positional constructor arguments and the reference's audio-net weights."""
    from idealnerf_amd.train_torso import Network
    from idealnerf_amd.helper import RenderConfig
    g = golden("head_torso")
    net = Network(32, 32, oracle.synthetic_frame(32, 32, seed=4)["focal"], NEAR, FAR, 512, 64, 128,
                  args=RenderConfig(perturb=0.0, chunk=512, near=NEAR, far=FAR, dim_expr=79)).to(dev)
    dh = oracle.facenerf_dims(dim_aud=64, dim_expr=79, dim_latent=32)
    dt = oracle.facenerf_dims(dim_aud=106, dim_expr=0, dim_latent=0)
    P = dict(hc=scale_sigma(oracle.xavier_facenerf_params(21, dh), 100.0, 0.2), hf=scale_sigma(oracle.xavier_facenerf_params(22, dh), 100.0, 0.2),
             tc=scale_sigma(oracle.xavier_facenerf_params(23, dt), 4.0, -0.2), tf=scale_sigma(oracle.xavier_facenerf_params(24, dt), 4.0, -0.2))
    net.face_nerf_coarse.load_state_dict(P["hc"]); net.face_nerf_fine.load_state_dict(P["hf"])
    net.torso_coarse_nerf.load_state_dict(P["tc"]); net.torso_fine_nerf.load_state_dict(P["tf"])
    net.aud_net.load_state_dict({k[len("audnet."):]: T(v) for k, v in g.items() if k.startswith("audnet.")})
    net.train()   # the reference ran its ray-batch branch (render_poses = None); autograd is off
    d = {k[3:]: T(v) for k, v in g.items() if k.startswith("in_")}
    x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
         d["expr"][None], d["latent"], torch.tensor([1]))
    with torch.no_grad():
        rgb_com, rgb_com0 = net([x, 0, 4])
    assert rel_err(rgb_com0, g["rgb_com0"]) < RGB_TOL              # no sampling before the coarse composite: 1e-4 outright
    # the two renders behind forward, with taps, against the reference's own parts
    head, torso, rays_h, rays_t, bg, (fold_h, fold_t) = _torso_hip_parts(net, d, dev)
    assert rel_err(rays_h, g["rays_head"]) < 1e-6 and rel_err(rays_t, g["rays_torso"]) < 1e-6
    assert torch.equal(head["rgb_map"] * torso["last_weight"][..., None] + torso["rgb_map_fg"], rgb_com)
    for tag, r in (("head", head), ("torso", torso)):
        for k in ("rgb0", "rgb_map_fg0", "last_weight0"):
            assert rel_err(r[k], g[f"{tag}_{k}"]) < RGB_TOL, (tag, k)
        check_stage(f"head+torso golden ({tag})", r, None, 128)     # the sampling stage is exact on the HIP coarse weights
    # the reference's merged fine depths: its own samples merged with the (bit-exact) coarse depths
    zf = lambda r, tag: torch.sort(torch.cat([r["tap_z_coarse"].cpu(), T(g[f"z_samples_{tag}"])], -1), -1)[0]
    fh, ft = net.face_nerf_fine, net.torso_fine_nerf
    compose = lambda h, t: h["rgb_map"] * t["last_weight"][..., None] + t["rgb_fg"]
    on_ref = compose(hip_fine_pass(idn, fh.packed_weights(), fold_h, rays_h, bg, zf(head, "head"), fh.prec_code),
                     hip_fine_pass(idn, ft.packed_weights(), fold_t, rays_t, bg, zf(torso, "torso"), ft.prec_code, with_fg=True))
    aud_feature = T(g["aud_feature"])
    with torch.no_grad():
        aud_torso = oracle.torso_signal(aud_feature, d["pose"])
    ora_on_hip = compose(oracle_fine_pass(P["hf"], dh, rays_h, bg, aud_feature, d["expr"], d["latent"], head["tap_z_fine"]),
                         oracle_fine_pass(P["tf"], dt, rays_t, bg, aud_torso, None, None, torso["tap_z_fine"], with_fg=True))
    (fl_h, rate_h), (fl_t, rate_t) = flipped_rows(head["tap_inds"], g["inds_head"]), flipped_rows(torso["tap_inds"], g["inds_torso"])
    same = ((zf(head, "head") == head["tap_z_fine"].cpu()).all(1) & (zf(torso, "torso") == torso["tap_z_fine"].cpu()).all(1)).numpy()
    res = prove("head+torso vs the reference's composite", (on_ref, g["rgb_com"]), (rgb_com, ora_on_hip), rgb_com, g["rgb_com"],
                fl_h | fl_t, max(rate_h, rate_t, key=float), FLIP_TOL_SHARP, same)
    assert res["beyond"] < 0.03 * 512     # a statistic of this scene (measured: 6 rays), not a criterion: the criteria are in prove()


def test_frame512_tile_golden(idn, dev, golden):
    """BASELINE configs[1] at FULL size against the reference itself (tests/golden/frame512_tile.npz: the reference's
    Network.render_rays on the first 4096 rays of the 512 x 512 bench frame): every output of every ray at the fixed
    1e-4 / 1e-5 budgets, importance indices against the reference's, and parity_proof's decomposition -- the fine pass on
    the reference's own sample positions (every 8th ray) and the oracle's fine pass on the HIP positions."""
    g = golden("frame512_tile")
    n = int(g["n_rays"])
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(512, 512, seed=0, dims=dims)
    cond = (syn["aud"], syn["expr"], syn["latent"])
    rays = idn.ops.frame_rays(syn["c2w"], 512, 512, syn["focal"], NEAR, FAR, row0=0, nrows=n // 512, device=dev)
    assert rel_err(rays[:4], g["rays_first"]) < 1e-6 and rel_err(rays[-4:], g["rays_last"]) < 1e-6
    bc = syn["bc"].reshape(-1, 3)[:n].contiguous().to(dev)
    fc, ff = fold_c(*cond), fold_f(*cond)
    out = idn.ops.render_rays_fwd(rays, bc, pk_c, fc, pk_f, ff, torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev),
                                  128, taps=True)
    for k in ("rgb0", "disp0", "acc0"):                       # nothing is sampled before the coarse composite
        assert rel_err(out[k], g[k]) < RGB_TOL, k
    check_stage("frame512 tile", out, None, 128)              # the sampling stage is exact on the HIP coarse weights
    fl, rate = flipped_rows(out["tap_inds"], g["inds"])
    sub = torch.arange(0, n, 8)
    sd = sub.to(dev)
    on_ref = hip_fine_pass(idn, pk_f, ff, rays[sd].contiguous(), bc[sd].contiguous(), T(g["z_fine_every8"]))
    # the oracle's fine pass on the HIP positions: every 8th ray and every ray that is beyond 1e-4 end to end
    e2e = np.abs(out["rgb_map"].cpu().numpy().astype(np.float64) - g["rgb_map"]).max(1) / np.abs(g["rgb_map"]).max()
    pick = torch.from_numpy(np.union1d(sub.numpy(), np.nonzero(e2e > RGB_TOL)[0]))
    pd = pick.to(dev)
    ora = oracle_fine_pass(pf, dims, rays[pd], bc[pd], *cond, out["tap_z_fine"][pd])
    for k in ("rgb_map", "disp_map", "acc_map"):
        _ = prove(f"frame512 tile {k}", (on_ref[k], g[k][sub.numpy()]), (out[k][pd], ora[k]), out[k], g[k], fl, rate, FLIP_TOL)
        if k == "rgb_map":
            stats32 = _
    keep = torch.from_numpy(~fl)
    assert abs_err(out["last_weight"][keep.to(dev)], g["last_weight"][keep.numpy()]) < W_TOL
    assert rel_err(out["z_std"][keep.to(dev)], g["z_std"][keep.numpy()]) < Z_STD_TOL
    # the same tile in the six-piece bf16 arithmetic, held to the same reference output: the default-precision criterion's third scene
    sd6 = [{k: v.to(dev).contiguous() for k, v in p_.items()} for p_ in (pc, pf)]
    ps6 = [idn.ops.params_struct(sd_, 64, 76, 32) for sd_ in sd6]
    pk6 = [idn.ops.pack_weights(ps_, dev, BF16X6_CODE) for ps_ in ps6]
    out6 = idn.ops.render_rays_fwd(rays, bc, pk6[0], fc, pk6[1], ff, torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev),
                                   128, taps=True, precision=BF16X6_CODE, precision_fine=BF16X6_CODE)
    check_stage("frame512 tile bf16x6", out6, None, 128)
    fl6, rate6 = flipped_rows(out6["tap_inds"], g["inds"])
    on_ref6 = hip_fine_pass(idn, pk6[1], ff, rays[sd].contiguous(), bc[sd].contiguous(), T(g["z_fine_every8"]), BF16X6_CODE)
    e2e6 = np.abs(out6["rgb_map"].cpu().numpy().astype(np.float64) - g["rgb_map"]).max(1) / np.abs(g["rgb_map"]).max()
    pick6 = torch.from_numpy(np.union1d(sub.numpy(), np.nonzero(e2e6 > RGB_TOL)[0]))
    pd6 = pick6.to(dev)
    ora6 = oracle_fine_pass(pf, dims, rays[pd6], bc[pd6], *cond, out6["tap_z_fine"][pd6])
    stats6 = prove("frame512 tile rgb_map (bf16x6)", (on_ref6["rgb_map"], g["rgb_map"][sub.numpy()]), (out6["rgb_map"][pd6], ora6["rgb_map"]),
                   out6["rgb_map"], g["rgb_map"], fl6, rate6, FLIP_TOL)
    assert_default_precision_allowed(default_precision_criterion("the reference's tile of the 512 x 512 frame", stats32, stats6, n),
                                     "the reference's tile of the 512 x 512 bench frame")


def test_torso_signal_golden(idn, dev, golden):
    """a11's conditioning against the reference's own pose_to_euler_trans (run_nerf_helpers.py:26-47) and
    signal assembly (train_torso.py:238-240, get_embedder(3, 0)), computed on the GPU by the product module."""
    from idealnerf_amd.helper import RenderConfig
    from idealnerf_amd.train_torso import Network, pose_to_euler_trans
    g = golden("torso_signal")
    net = Network(8, 8, 100.0, NEAR, FAR, 64, 64, 128, args=RenderConfig(dim_expr=79)).to(dev)
    poses, aud = T(g["poses"]).to(dev), T(g["aud"]).to(dev)
    et = pose_to_euler_trans(poses)
    assert abs_err(et, g["euler_trans"]) < 2e-6          # atan2 / asin: CPU SLEEF vs GPU ocml, last ulps
    for b in range(poses.shape[0]):
        sig = net.torso_signal(aud[b], poses[b])
        assert sig.shape == (106,) and sig.device.type == "cuda"
        np.testing.assert_array_equal(sig[:64].cpu().numpy(), g["signal"][b, :64])
        assert abs_err(sig, g["signal"][b]) < 1e-5        # sin/cos of up to 4x the angle


def test_head_torso_gradients_match_oracle(idn, dev):
    net, syn, P, dims, d = _torso_setup(idn, dev)
    net.train()
    x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
         d["expr"][None], d["latent"], torch.tensor([1]))
    rgb_com, rgb_com0 = net([x, 0, 4])
    tgt = d["target"].to(dev)
    loss = ((rgb_com - tgt) ** 2).mean() + ((rgb_com0 - tgt) ** 2).mean()
    loss.backward()
    for p in P.values():
        for v in p.values():
            v.requires_grad_(True)
    (ref, ref0), aud_net = _torso_oracle(net, P, dims, d, grad=True)
    loss_o = ((ref - d["target"]) ** 2).mean() + ((ref0 - d["target"]) ** 2).mean()
    loss_o.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-5 * abs(float(loss_o))
    pairs = (("hc", net.face_nerf_coarse), ("tc", net.torso_coarse_nerf), ("hf", net.face_nerf_fine), ("tf", net.torso_fine_nerf))
    for tag, m in pairs:
        check_grads(m.named_parameters(), P[tag], tag.endswith("f"), tag)
    # the audio net is reached through d aud of both pairs (torso: only the first 64 channels)
    for (name, prm), (_, ref_p) in zip(net.aud_net.named_parameters(), aud_net.named_parameters()):
        assert l2_err(prm.grad, ref_p.grad) < GRAD_L2_FINE, name


def test_train_loop_adam_steps_match_oracle(idn, dev):
    """Three iterations of the reference's loop body (audio_exp_nerf.py:530-558) through
    Network.forward's 9-tuple: loss trajectory, PSNR, learning rate and the updated weights
    against the CPU oracle driven by the same torch Adam."""
    from idealnerf_amd import train as T_
    net, syn = _train_net(idn, dev)
    dims = oracle.facenerf_dims()
    rs = np.random.RandomState(3)
    n = 96
    ro, rd = oracle.camera_rays(32, 32, syn["focal"], syn["c2w"])
    sel = T(rs.choice(1024, size=n, replace=False))
    batch_rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    bg = syn["bc"].reshape(-1, 3)[sel].contiguous()
    tgt = T(rs.uniform(0, 1, size=(n, 3)).astype(np.float32))
    auds = T(rs.standard_normal((4, 16, 29)).astype(np.float32))
    raw_img = torch.zeros(1, 32, 32, 3)
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
    latent_codes = torch.ones(4, 32, device=dev, requires_grad=True)
    # oracle twin on the CPU
    cpu = lambda m: {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    pc, pf = cpu(net.face_nerf_coarse), cpu(net.face_nerf_fine)
    aud_net_o = type(net.aud_net)(64, 16)
    aud_net_o.load_state_dict({k: v.detach().cpu() for k, v in net.aud_net.state_dict().items()})
    lat_o = torch.ones(4, 32, requires_grad=True)
    live = lambda p: [v for k, v in p.items() if not k.startswith("feature_linear")]
    opt_o = torch.optim.Adam(live(pc) + live(pf) + list(aud_net_o.parameters()) + [lat_o], lr=8e-4, betas=(0.9, 0.999))
    opt = T_.make_optimizer(net, latent_codes, lrate=8e-4)
    rays_rec = oracle.ray_records(batch_rays[0], batch_rays[1], NEAR, FAR)
    data = (batch_rays[None], tgt, bg, auds[None], raw_img, pose, syn["expr"][None], torch.tensor([2]))
    for step in range(3):
        info = T_.train_step(net, opt, data, latent_codes, step, 4, lrate=8e-4, lrate_decay=500)
        opt_o.zero_grad()
        aud_f = aud_net_o(auds[2:3])
        out = oracle.render_rays(rays_rec, bg, pc, pf, aud_f, syn["expr"], lat_o[2], dims=dims)
        loss_o, img_o = oracle.train_loss(out, tgt, lat_o[2])
        loss_o.backward()
        opt_o.step()
        lr_o = 8e-4 * (0.1 ** (step / (500 * 1500)))
        for gq in opt_o.param_groups:
            gq["lr"] = lr_o
        assert abs(float(info["loss"]) - float(loss_o)) < 2e-4 * abs(float(loss_o)), step
        assert abs(float(info["psnr"]) - float(oracle.mse_to_psnr(img_o.detach()))) < 1e-2
        assert info["lr"] == pytest.approx(lr_o)
    for name in ("pts_linears.0.weight", "pts_linears.5.weight", "views_linears.0.weight", "alpha_linear.weight",
                 "rgb_linear.bias"):
        got = dict(net.face_nerf_fine.named_parameters())[name].detach().cpu()
        # Adam's first steps move every weight by ~lr * sign(g): entries whose gradient is
        # rounding noise around zero can land 2*lr apart, everything else must agree
        diff = (got - pf[name].detach()).abs()
        assert float((diff > 1e-4).float().mean()) < 0.02, name
        assert float(diff.median()) < 1e-5, name
        assert float(diff.max()) < 3 * 2 * 8e-4 + 1e-6, name
    assert abs_err(latent_codes[2], lat_o[2]) < 2e-4
    assert torch.equal(latent_codes[0].detach().cpu(), torch.ones(32))  # untouched frames keep their code
    assert dict(net.face_nerf_fine.named_parameters())["feature_linear.weight"].grad is None


def test_backward_kernels_vs_fp64_on_saved_activations(idn, dev):
    """The arithmetic of the backward pass (compositing backward, delta/dW GEMMs, bias sums,
    conditioning fold) against an fp64 torch backprop that uses the SAME saved activations, hence
    the same ReLU masks: every gradient tensor, including the folded conditioning columns."""
    from idealnerf_amd import autograd as ag
    from idealnerf_amd.helper import linspace01
    net, syn, P, dims, d = _torso_setup(idn, dev)
    for coarse_net, S in ((net.face_nerf_coarse, 64), (net.face_nerf_fine, 192)):
        with torch.no_grad():
            aud = net.aud_net(d["auds"][1:2].to(dev)).contiguous()
        expr, lat = d["expr"].to(dev), d["latent"].to(dev)
        rec = oracle.ray_records(d["batch_rays"][0], d["batch_rays"][1], NEAR, FAR).to(dev)
        bc = d["bg"].to(dev).contiguous()
        z = idn.ops.coarse_depths(rec, linspace01(S, dev))
        folded = coarse_net.folded_bias(aud, expr, lat)
        raw, acts = ag._train_query(coarse_net, folded, rec, z)
        n = rec.shape[0]
        Pn = n * S
        Pp = (Pn + 127) // 128 * 128
        rs = np.random.RandomState(S)
        g_rgb = T(rs.standard_normal((n, 3)).astype(np.float32) * 0.01).to(dev)
        g_lw = T(rs.standard_normal(n).astype(np.float32) * 0.01).to(dev)
        d_aud, d_lat = torch.zeros_like(aud), torch.zeros_like(lat)
        grads = ag._pass_bwd(coarse_net, aud, expr, lat, acts, raw, z, rec, bc, g_rgb, None, g_lw, None, d_aud, d_lat)
        mat = lambda o, w: acts[o * Pp:(o + w) * Pp].view(Pp, w)[:Pn].double()
        x0, dirs = mat(0, 64), mat(64, 64)
        a = [mat(128 + 256 * i, 256) for i in range(8)]
        v = [mat(128 + 2048 + 128 * i, 128) for i in range(3)]
        raw64 = raw.double().cpu().requires_grad_(True)
        comp = oracle.composite(raw64, z.double().cpu(), rec[:, 3:6].double().cpu(), bc.double().cpu())
        ((comp[0] * g_rgb.double().cpu()).sum() + (comp[3][:, -1] * g_lw.double().cpu()).sum()).backward()
        d_raw = raw64.grad.to(dev).view(Pn, 4)
        sd = {k: p.detach().double() for k, p in coarse_net.named_parameters()}
        C = 64 + 79 + 32
        cond = torch.cat([aud.double(), expr.double() / 3, lat.double()])
        G = {}
        d_rgb, d_sig = d_raw[:, :3], d_raw[:, 3:4]
        G["rgb_linear.weight"], G["rgb_linear.bias"] = d_rgb.t() @ v[2], d_rgb.sum(0)
        dl = (d_rgb @ sd["rgb_linear.weight"]) * (v[2] > 0)
        for i in (2, 1):
            G[f"views_linears.{i}.weight"], G[f"views_linears.{i}.bias"] = dl.t() @ v[i - 1], dl.sum(0)
            dl = (dl @ sd[f"views_linears.{i}.weight"]) * (v[i - 1] > 0)
        inp_v0 = torch.cat([a[7], dirs[:, :27], (expr.double() / 3)[None].expand(Pn, -1)], 1)
        G["views_linears.0.weight"], G["views_linears.0.bias"] = dl.t() @ inp_v0, dl.sum(0)
        G["alpha_linear.weight"], G["alpha_linear.bias"] = d_sig.t() @ a[7], d_sig.sum(0)
        dh = (dl @ sd["views_linears.0.weight"][:, :256] + d_sig @ sd["alpha_linear.weight"]) * (a[7] > 0)
        d_cond = torch.zeros(C, dtype=torch.float64, device=dev)
        for l in range(7, 0, -1):
            inp = a[l - 1] if l != 5 else torch.cat([x0[:, :63], cond[None].expand(Pn, -1), a[4]], 1)
            G[f"pts_linears.{l}.weight"], G[f"pts_linears.{l}.bias"] = dh.t() @ inp, dh.sum(0)
            W = sd[f"pts_linears.{l}.weight"]
            if l == 5:
                d_cond += (dh @ W[:, 63:63 + C]).sum(0)
                W = W[:, 63 + C:]
            dh = (dh @ W) * (a[l - 1] > 0)
        inp0 = torch.cat([x0[:, :63], cond[None].expand(Pn, -1)], 1)
        G["pts_linears.0.weight"], G["pts_linears.0.bias"] = dh.t() @ inp0, dh.sum(0)
        d_cond += (dh @ sd["pts_linears.0.weight"][:, 63:]).sum(0)
        for k in G:
            assert rel_err(grads[k], G[k]) < 5e-6, (S, k)
        assert rel_err(d_aud, d_cond[:64]) < 5e-6 and rel_err(d_lat, d_cond[64 + 79:]) < 5e-6


def test_dw_gemm_bf16_pieces_match_the_fp32_pipe_against_fp64(idn, dev):
    """The 256 x 256 weight-gradient GEMM of the training step runs on the bf16 matrix pipe: every fp32 operand as the
    exact sum of three bf16 pieces, six piece products per product, fp32 accumulate.  Its result must be fp32-grade:
    compared with an fp64 product of the same inputs it may not be further off than the fp32-MFMA kernel by more than a
    small factor, on inputs a training step does not produce -- magnitudes over 24 octaves (pieces that underflow each
    other), heavy cancellation, post-ReLU zeros, one split with a single 16-row chunk, row pitches wider than 256."""
    rs = np.random.RandomState(11)
    worst = 0.0
    for rows, pitch_d, pitch_a, kind in ((128, 256, 256, "normal"), (4096, 256, 320, "wide"), (20480, 384, 256, "wide"),
                                         (12800, 256, 256, "cancel"), (6400, 256, 256, "relu")):
        d = rs.standard_normal((rows, pitch_d)).astype(np.float32)
        a = rs.standard_normal((rows, pitch_a)).astype(np.float32)
        if kind == "wide":      # log-uniform magnitudes, 2^-12 .. 2^12, per element
            d *= np.exp2(rs.uniform(-12, 12, d.shape)).astype(np.float32)
            a *= np.exp2(rs.uniform(-12, 12, a.shape)).astype(np.float32)
        elif kind == "cancel":  # every column's sum cancels to ~1e-4 of its terms
            d[1::2] = -d[0::2] * (1 + 1e-4 * rs.standard_normal(d[0::2].shape).astype(np.float32))
            a[1::2] = a[0::2]
        elif kind == "relu":
            a = np.maximum(a, 0)
            d *= (rs.uniform(size=d.shape) < 0.1)   # sparse deltas
        dt, at = T(d).to(dev), T(a).to(dev)
        ref = dt[:, :256].double().t() @ at[:, :256].double()
        scale = dt[:, :256].double().abs().t() @ at[:, :256].double().abs()   # sum |a||b| per output: the error unit
        ref_b = dt[:, :256].double().sum(0)
        err = {}
        for pipe in (0, 1):
            dW, db = idn.ops.dw_gemm(dt, at, pipe)
            assert torch.isfinite(dW).all()
            err[pipe] = float(((dW.double() - ref).abs() / scale).max())
            eb = float((db.double() - ref_b).abs().max() / dt[:, :256].double().abs().sum(0).max())
            assert eb < 3e-7, (kind, rows, pipe, eb)
        print(f"\n  dW GEMM {kind:6s} rows {rows:6d}: max |err| / sum|a||b|: bf16 pieces {err[0]:.2e}, fp32 pipe {err[1]:.2e}")
        assert err[0] < 2e-7 and err[0] < 4 * err[1] + 3e-8, (kind, rows, err)
        worst = max(worst, err[0])
    # bit-reproducible (no atomics, fixed split and reduction order)
    dW1, _ = idn.ops.dw_gemm(dt, at, 0)
    dW2, _ = idn.ops.dw_gemm(dt, at, 0)
    assert torch.equal(dW1, dW2)
    with pytest.raises(RuntimeError):
        idn.ops.dw_gemm(dt[:100].contiguous(), at[:100].contiguous(), 0)   # rows not a multiple of 128
    # At bench scale, with the split count a training pass uses (pipe 2: one product per workgroup, 2 #CUs / 9 workgroups per
    # product -- each keeps its fp32 accumulators over 10 500 of the fine pass's 589 824 points, against 2 304 with a split per
    # CU): post-ReLU activations and sparse deltas as a step produces them.  fp32-grade means here what it means for the
    # reference's own GEMM: products to 2^-23, sums carried in fp32 over a block of the points and blocks added in fp64 -- measured
    # next to torch's fp32 matmul of the same operands (one fp32 GEMM over all rows: what the reference's autograd runs).
    rows = 3072 * 192
    g = torch.Generator(device=dev).manual_seed(5)
    at = torch.relu(torch.randn((rows, 256), generator=g, device=dev))
    dt = torch.randn((rows, 256), generator=g, device=dev) * (torch.rand((rows, 256), generator=g, device=dev) < 0.5)
    ref, scale = torch.zeros((256, 256), dtype=torch.float64, device=dev), torch.zeros((256, 256), dtype=torch.float64, device=dev)
    for r0 in range(0, rows, 49152):     # fp64 reference in slabs (the whole product in fp64 would hold 2.4 GB of operands)
        d64, a64 = dt[r0:r0 + 49152].double(), at[r0:r0 + 49152].double()
        ref += d64.t() @ a64
        scale += d64.abs().t() @ a64.abs()
    err = {}
    both = lambda dW: (float(((dW.double() - ref).abs() / scale).max()), float(((dW.double() - ref).abs().max() / ref.abs().max())))
    for pipe in (2, 0, 1):
        err[pipe] = both(idn.ops.dw_gemm(dt, at, pipe)[0])
    err["torch"] = both(dt.t() @ at)
    print(f"  dW GEMM at bench scale ({rows} rows): max |err| / sum|a||b| (and / max |dW|): one product per workgroup {err[2][0]:.2e} ({err[2][1]:.2e}), "
          f"a split per CU {err[0][0]:.2e} ({err[0][1]:.2e}), fp32 pipe {err[1][0]:.2e} ({err[1][1]:.2e}), torch fp32 matmul {err['torch'][0]:.2e} ({err['torch'][1]:.2e})")
    assert err[2][0] < 1e-7 and err[2][1] < 3e-6, err
    dW1, _ = idn.ops.dw_gemm(dt, at, 2)
    assert torch.equal(dW1, idn.ops.dw_gemm(dt, at, 2)[0])


# --------------------------------------------------------------------------- bf16x3 arithmetic mode
BF16X3 = 1  # IDN_PREC_BF16X3


def test_bf16x3_facenerf_golden(idn, dev, golden):
    """Three bf16 MFMAs per product: ~2^-16 per product, ~1.5e-5 on the output."""
    g = golden("facenerf")
    for name, v in (("c235", dict(dim_aud=64, dim_expr=76, dim_latent=32)), ("c169", dict(dim_aud=106, dim_expr=0, dim_latent=0))):
        dims = oracle.facenerf_dims(**v)
        params = oracle.xavier_facenerf_params(11, dims)
        sd = {k: t.to(dev).contiguous() for k, t in params.items()}
        ps = idn.ops.params_struct(sd, dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])
        packed = idn.ops.pack_weights(ps, dev, BF16X3)
        opt = lambda k: T(g[k]).to(dev) if k in g else None
        folded = idn.ops.fold_conditioning(ps, T(g[name + "_aud"]).to(dev), opt(name + "_expr"), opt(name + "_latent"), dev)
        out = idn.ops.facenerf_fwd(packed, folded, T(g[name + "_x"]).to(dev), BF16X3)
        err = rel_err(out, g[name + "_out"])
        print(f"\nbf16x3 FaceNeRF {name}: max rel err vs reference = {err:.2e}")
        assert err < 5e-5


@pytest.mark.parametrize("n", [1, 33, 130, 4099])
def test_bf16x3_ragged(idn, dev, n):
    dims = oracle.facenerf_dims()
    params = scale_sigma(oracle.xavier_facenerf_params(5, dims), 30.0, 0.1)
    rs = np.random.RandomState(n)
    x = T(rs.uniform(-1, 1, size=(n, 90)).astype(np.float32))
    aud, expr, lat = (T(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
    with torch.no_grad():
        ref = oracle.facenerf_forward(params, x, aud, expr, lat, dims)
    sd = {k: t.to(dev).contiguous() for k, t in params.items()}
    ps = idn.ops.params_struct(sd, 64, 76, 32)
    out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, BF16X3),
                               idn.ops.fold_conditioning(ps, aud.to(dev), expr.to(dev), lat.to(dev), dev), x.to(dev), BF16X3)
    assert rel_err(out, ref) < 1e-4


def test_bf16x3_render_frame32_golden(idn, dev, golden):
    """End to end in bf16x3: RGB within the 1e-4 budget, PSNR vs the reference's frame."""
    g = golden("frame32")
    dims = oracle.facenerf_dims()
    pc = scale_sigma(oracle.xavier_facenerf_params(2, dims))
    pf = scale_sigma(oracle.xavier_facenerf_params(3, dims))
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    cond = [t.to(dev) for t in (syn["aud"], syn["expr"], syn["latent"])]
    nets = []
    for p in (pc, pf):
        sd = {k: t.to(dev).contiguous() for k, t in p.items()}
        ps = idn.ops.params_struct(sd, 64, 76, 32)
        nets.append((idn.ops.pack_weights(ps, dev, BF16X3), idn.ops.fold_conditioning(ps, *cond, dev), sd))
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    out = idn.ops.render_rays_fwd(rays, syn["bc"].reshape(-1, 3).to(dev), nets[0][0], nets[0][1], nets[1][0], nets[1][1],
                                  torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev), 128,
                                  taps=True, precision=BF16X3)
    e_rgb, e_rgb0 = rel_err(out["rgb_map"], g["rgb"].reshape(-1, 3)), rel_err(out["rgb0"], g["rgb0"].reshape(-1, 3))
    flips = (out["tap_inds"].cpu().numpy() != g["tap_inds"].astype(np.int64)).mean()
    mse = float(((out["rgb_map"].cpu().numpy() - g["rgb"].reshape(-1, 3)) ** 2).mean())
    print(f"\nbf16x3 frame32: rgb err {e_rgb:.2e}, rgb0 err {e_rgb0:.2e}, index flip rate {flips:.2e}, "
          f"PSNR vs reference {10 * np.log10(1.0 / max(mse, 1e-30)):.1f} dB")
    assert e_rgb < RGB_TOL and e_rgb0 < RGB_TOL
    assert flips < FLIP_TOL_X3
    assert rel_err(out["tap_raw_coarse"][:128], g["tap_raw_coarse"]) < 1e-4


def test_bf16x3_module_precision_switch(idn, dev, golden):
    g = golden("facenerf")
    net = idn.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76).to(dev)
    net.load_state_dict(oracle.xavier_facenerf_params(11, oracle.facenerf_dims()))
    args = [T(g["c235_" + k]).to(dev) for k in ("x", "aud", "expr", "latent")]
    with torch.no_grad():
        o32 = net(*args)
        net.precision = "bf16x3"
        o16 = net(*args)
    assert rel_err(o32, g["c235_out"]) < 1e-5 and rel_err(o16, g["c235_out"]) < 5e-5
    assert not torch.equal(o32, o16)


# --------------------------------------------------------------------------- plain bf16 (config 5)
BF16 = 2  # IDN_PREC_BF16
# measured (profiles/r04_pytest_gpu_*.log): the median point is 2.4e-8 from the rounding model -- the accumulation order is all
# that differs --, 1-2 % of the points carry an activation that sat on a bf16 rounding boundary and moved by 2^-8 of itself
# (1e-4 .. 6e-3 on the output, the size of the arithmetic's own distance from fp32)
BF16_MODEL_TYPICAL = 1e-6   # median point vs the rounding model
BF16_MODEL_FLIPPED = 0.10   # share of the points that may be beyond 1e-4 (rounding-boundary flips; measured 1-2 %)
BF16_MODEL_WORST = 2e-2     # no point at all beyond this


def _bf16_vs_emulation(idn, dev, params, x, aud, expr, lat, dims, what):
    """The plain-bf16 kernel against the oracle's model of its rounding (oracle.facenerf_forward_bf16_emulated: weights and
    per-point layer inputs rounded to bf16 once, exact products, fp32 folded biases).  Errors are relative to max |ref|.
    The bound is what separates "the arithmetic the model states" from anything else: a hazard that feeds stale operands
    to 2 of 16 channels (shipped in rounds 1-2 under a 3e-2 bound) moves EVERY point by ~1e-2; what the model cannot pin
    is an activation within the accumulation noise of a bf16 rounding boundary, which then moves by one bf16 ulp
    (2^-8 of itself) at a rare point: hence a very tight bound on the typical point (1e-6: measured 2e-8), a bound on
    the SHARE of points that carry such a flip, and a loose one on the worst point."""
    cond = [None if t is None else t.to(dev) for t in (aud, expr, lat)]
    with torch.no_grad():
        emu = oracle.facenerf_forward_bf16_emulated(params, x, aud, expr, lat, dims).double()
        f32 = oracle.facenerf_forward(params, x, aud, expr, lat, dims).double()
    sd = {k: t.to(dev).contiguous() for k, t in params.items()}
    ps = idn.ops.params_struct(sd, dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])
    out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, BF16), idn.ops.fold_conditioning(ps, *cond, dev), x.to(dev), BF16)
    assert out.shape == emu.shape
    scale = float(emu.abs().max())
    e = ((out.cpu().double() - emu).abs().max(1)[0] / scale).numpy()
    vs32 = float((out.cpu().double() - f32).abs().max() / f32.abs().max())
    print(f"\n  plain bf16 {what}: vs the bf16 rounding model median {np.median(e):.2e}, 99th pct {np.percentile(e, 99):.2e}, "
          f"max {e.max():.2e} ({len(e)} points); vs fp32 (report) {vs32:.2e}")
    return e, vs32


def test_bf16_facenerf_follows_its_rounding_model(idn, dev, golden):
    """Plain bf16 operands, fp32 accumulate (BASELINE configs[4]): ~6.5e-3 from the fp32 reference output by design --
    and within ~1e-5 of what rounding weights and layer inputs to bf16 PREDICTS, on the reference's golden inputs."""
    g = golden("facenerf")
    for name, v in (("c235", dict(dim_aud=64, dim_expr=76, dim_latent=32)), ("c169", dict(dim_aud=106, dim_expr=0, dim_latent=0))):
        dims = oracle.facenerf_dims(**v)
        params = oracle.xavier_facenerf_params(11, dims)
        opt = lambda k: T(g[f"{name}_{k}"]) if f"{name}_{k}" in g and g[f"{name}_{k}"].size else None
        e, vs32 = _bf16_vs_emulation(idn, dev, params, T(g[f"{name}_x"]), opt("aud"), opt("expr"), opt("latent"), dims, name)
        assert np.median(e) < BF16_MODEL_TYPICAL and (e > 1e-4).mean() < BF16_MODEL_FLIPPED and e.max() < BF16_MODEL_WORST, \
            (name, np.median(e), (e > 1e-4).mean(), e.max())
        assert 1e-4 < vs32 < 3e-2      # it IS bf16: outside the 1e-4 budget, offered for the PSNR-judged config only


@pytest.mark.parametrize("n", [1, 130, 4099])
def test_bf16_ragged(idn, dev, n):
    dims = oracle.facenerf_dims()
    params = scale_sigma(oracle.xavier_facenerf_params(5, dims), 30.0, 0.1)
    rs = np.random.RandomState(n)
    x = T(rs.uniform(-1, 1, size=(n, 90)).astype(np.float32))
    aud, expr, lat = (T(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
    e, vs32 = _bf16_vs_emulation(idn, dev, params, x, aud, expr, lat, dims, f"ragged n={n}")
    assert np.median(e) < BF16_MODEL_TYPICAL and (e > 1e-4).mean() < max(BF16_MODEL_FLIPPED, 1.5 / n) and e.max() < BF16_MODEL_WORST and vs32 < 3e-2


def test_bf16_render_frame32_psnr(idn, dev, golden):
    """BASELINE config 5's criterion: PSNR against a target image within 0.05 dB of the
    reference's own PSNR against it.  Target = the reference's frame plus 1/255-level noise
    (a ~48 dB fit, above what trained models reach, so the criterion is at its strictest)."""
    g = golden("frame32")
    dims = oracle.facenerf_dims()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    cond = [t.to(dev) for t in (syn["aud"], syn["expr"], syn["latent"])]
    nets = []
    for seed in (2, 3):
        sd = {k: t.to(dev).contiguous() for k, t in scale_sigma(oracle.xavier_facenerf_params(seed, dims)).items()}
        ps = idn.ops.params_struct(sd, 64, 76, 32)
        nets.append((idn.ops.pack_weights(ps, dev, BF16), idn.ops.fold_conditioning(ps, *cond, dev), sd))
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    out = idn.ops.render_rays_fwd(rays, syn["bc"].reshape(-1, 3).to(dev), nets[0][0], nets[0][1], nets[1][0], nets[1][1],
                                  torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev), 128,
                                  precision=BF16)
    ours, ref = out["rgb_map"].cpu().numpy().astype(np.float64), g["rgb"].reshape(-1, 3).astype(np.float64)
    target = ref + np.random.RandomState(0).normal(0.0, 1.0 / 255.0, ref.shape)
    psnr = lambda a: -10.0 * np.log10(((a - target) ** 2).mean())
    direct = -10.0 * np.log10(max(((ours - ref) ** 2).mean(), 1e-30))
    print(f"\nplain bf16 frame32: PSNR vs reference frame {direct:.1f} dB; vs target: ours {psnr(ours):.3f} dB, "
          f"reference {psnr(ref):.3f} dB")
    assert abs(psnr(ours) - psnr(ref)) < 0.05
    assert direct > 45.0


def test_bf16_module_eval(idn, dev):
    """The module-level switch reaches the plain-bf16 kernel (inference only; training stays fp32)."""
    net = idn.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76).to(dev)
    net.precision = "bf16"
    x = torch.zeros(8, 90, device=dev)
    with torch.no_grad():
        assert net(x, torch.zeros(64, device=dev), torch.zeros(76, device=dev), torch.zeros(32, device=dev)).shape == (8, 4)


def test_dataset_sample_rays_golden(idn, dev, golden):
    """GetData.sample_rays on the device against the reference run with the same numpy seed."""
    from idealnerf_amd.dataset import sample_rays
    g = golden("sample_rays")
    H, W = g["parse"].shape[:2]
    np.random.seed(int(g["seed"]))
    br, ts, bs = sample_rays(g["pose"], g["rect"], T(g["target"]).to(dev), T(g["bc"]).to(dev), g["landmark"], g["parse"],
                             H, W, float(g["focal"]), float(g["cx"]), float(g["cy"]), int(g["N_rand"]),
                             int(g["mouth_rays"]), int(g["torso_rays"]), float(g["sample_rate"]), dev)
    assert br.shape == (2, 96, 3)
    assert rel_err(br, g["batch_rays"]) < 1e-6
    np.testing.assert_array_equal(ts.cpu().numpy(), g["target_s"])
    np.testing.assert_array_equal(bs.cpu().numpy(), g["bc_s"])


# --------------------------------------------------------------------------- 8(f) items 3 and 4
def test_to8b_bit_exact_and_nonfinite_flag(idn, dev):
    """helper.py:154 `(255 * np.clip(x, 0, 1)).astype(np.uint8)` on the device, byte for byte."""
    rs = np.random.RandomState(5)
    x = rs.uniform(-0.2, 1.2, size=(257, 33, 3)).astype(np.float32)
    x.reshape(-1)[:256] = np.arange(256, dtype=np.float32) / np.float32(255.0)   # the exact k/255 grid
    x.reshape(-1)[256:260] = [0.0, 1.0, -0.0, np.nextafter(np.float32(1.0), np.float32(0.0))]
    ref = (255 * np.clip(x, 0, 1)).astype(np.uint8)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    out = idn.ops.to8b(T(x).to(dev), False, flag)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    np.testing.assert_array_equal(idn.ops.to8b(T(x).to(dev), True).cpu().numpy(), ref[..., ::-1])
    assert int(flag.item()) == 0
    x[100, 7, 1], x[3, 3, 0] = np.nan, np.inf
    out = idn.ops.to8b(T(x).to(dev), False, flag).cpu().numpy()
    assert int(flag.item()) == 1 and out[100, 7, 1] == 0 and out[3, 3, 0] == 255
    mask = np.isfinite(x)
    np.testing.assert_array_equal(out[mask], ref[mask])


def test_frame_sink_overlapped_copies(idn, dev, tmp_path):
    """Frames submitted back to back come out in order, byte-identical to the host formula, and the
    frame holding a NaN is the one reported."""
    from idealnerf_amd.frame_io import FrameSink
    H, W, N = 48, 40, 5
    rs = np.random.RandomState(9)
    frames = [rs.uniform(-0.1, 1.1, size=(H * W, 3)).astype(np.float32) for _ in range(N)]
    frames[3][17, 2] = np.nan
    sink = FrameSink(str(tmp_path / "clip.avi"), W, H, fps=25.0, device=dev, keep_frames=True, codec="raw")
    for f in frames:
        sink.submit(T(f).to(dev))
    sink.release()
    assert sink.nonfinite_frames == [3] and len(sink.frames) == N
    for got, f in zip(sink.frames, frames):
        ref = (255 * np.clip(np.nan_to_num(f, nan=0.0), 0, 1)).astype(np.uint8).reshape(H, W, 3)
        np.testing.assert_array_equal(got, ref)
    assert (tmp_path / "clip.avi").stat().st_size > N * H * W * 3


def test_frame_sink_mjpg_clip_and_stills(idn, dev, tmp_path):
    """The reference's output (eval_aud_exp_nerf.py:482-496): an MJPG AVI at 25 fps plus a JPEG every 10th
    frame.  Each '00dc' chunk, found through the idx1 index, decodes to the to8b frame within JPEG error."""
    import io

    from PIL import Image
    from idealnerf_amd.frame_io import FrameSink, read_avi_chunks
    H, W, N = 64, 80, 12
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    frames = [np.stack([0.5 + 0.5 * np.sin(6 * xx + 0.3 * i), yy, 0.5 + 0.5 * np.cos(5 * yy * xx + 0.2 * i)], -1)
              .astype(np.float32).reshape(-1, 3) for i in range(N)]
    sink = FrameSink(str(tmp_path / "clip.avi"), W, H, fps=25.0, device=dev, keep_frames=True,
                     still_every=10, still_path=str(tmp_path / "still_{i}.jpg"))
    for f in frames:
        sink.submit(T(f).to(dev))
    sink.release()
    info, chunks = read_avi_chunks(str(tmp_path / "clip.avi"))
    assert info["handler"] == b"MJPG" and info["compression"] == b"MJPG" and info["chunk_id"] == b"00dc"
    assert (info["width"], info["height"], info["frames"]) == (W, H, N) and abs(info["fps"] - 25.0) < 1e-6
    assert len(chunks) == N
    for got8, chunk in zip(sink.frames, chunks):
        assert chunk[:2] == b"\xff\xd8"                        # a JPEG per chunk
        dec = np.asarray(Image.open(io.BytesIO(chunk)).convert("RGB"))[..., ::-1]   # back to the BGR order handed in
        assert dec.shape == (H, W, 3)
        assert np.abs(dec.astype(np.int32) - got8.astype(np.int32)).mean() < 2.0    # smooth content at quality 95
    assert sorted(sink.stills) == [str(tmp_path / "still_0.jpg"), str(tmp_path / "still_10.jpg")]
    still = np.asarray(Image.open(sink.stills[0]).convert("RGB"))[..., ::-1]
    assert np.abs(still.astype(np.int32) - sink.frames[0].astype(np.int32)).mean() < 2.0


def _agg_state(seed):
    import collections
    shapes = collections.OrderedDict([("agg_linears.0", (64, 140)), ("agg_linears.1", (64, 64))])
    c_all = 63 + 64 + 32
    for i in range(8):
        shapes[f"pts_linears.{i}"] = (256, c_all if i == 0 else (256 + c_all if i == 5 else 256))
    shapes.update([("views_linears.0", (128, 27 + 256 + 64)), ("views_linears.1", (128, 128)), ("views_linears.2", (128, 128)),
                   ("feature_linear", (256, 256)), ("alpha_linear", (1, 256)), ("rgb_linear", (3, 128))])
    rs = np.random.RandomState(seed)
    sd = collections.OrderedDict()
    for k, (o, i) in shapes.items():
        bound = float(np.sqrt(6.0 / (o + i)))
        sd[k + ".weight"] = T(rs.uniform(-bound, bound, size=(o, i)).astype(np.float32))
        sd[k + ".bias"] = torch.full((o,), 0.01, dtype=torch.float32)
    return sd


def test_facenerf_agg_golden(idn, dev, golden):
    """FaceNeRFAgg (models/face_nerf_agg.py) on the same fused kernel, against the reference module."""
    g = golden("facenerf_agg")
    net = idn.FaceNeRFAgg(dim_agg=64, dim_aud=64, dim_expr=76, dim_latent=32)
    sd = _agg_state(int(g["seed"]))
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net = net.to(dev)
    args = [T(g[k]).to(dev) for k in ("x", "aud", "expr", "latent")]
    with torch.no_grad():
        out = net(*args)
        net.precision = "bf16x3"
        out3 = net(*args)
    assert rel_err(out, g["out"]) < 1e-5 and rel_err(out3, g["out"]) < 5e-5
    with pytest.raises(NotImplementedError):
        net(*args)


def test_concurrent_host_threads_on_two_streams(idn, dev):
    """under nn.DataParallel the reference calls forward from one host thread per replica.
    Two threads, each on its own stream with its own network, must get what a serial call gets
    (no shared mutable state in the library besides the per-device launch setup)."""
    import threading
    dims = oracle.facenerf_dims()
    nets, xs, refs = [], [], []
    rs = np.random.RandomState(3)
    for seed in (21, 22):
        net = idn.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76)
        net.load_state_dict(oracle.xavier_facenerf_params(seed, dims))
        net = net.to(dev)
        net.precision = "bf16x3" if seed == 22 else "f32"
        x = T(rs.uniform(-1, 1, size=(20000, 90)).astype(np.float32)).to(dev)
        cond = [T(rs.standard_normal(k).astype(np.float32)).to(dev) for k in (64, 76, 32)]
        with torch.no_grad():
            refs.append(net(x, *cond).clone())
        nets.append(net); xs.append((x, cond))
    torch.cuda.synchronize()
    outs, errs = [None, None], []

    def work(i):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(5):
                    o = nets[i](xs[i][0], *xs[i][1])
                s.synchronize()
            outs[i] = o
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)


@pytest.mark.parametrize("n_rays,S,Ni", [(37, 40, 72), (5, 17, 33), (130, 96, 160)])
def test_render_unusual_sample_counts_vs_oracle(idn, dev, n_rays, S, Ni):
    """Sample counts other than the paper's 64/128 and ragged ray counts, all three arithmetic
    modes within their own budgets against the CPU oracle (which is pinned to the reference at 64/128)."""
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(32, 32, seed=4, dims=dims)
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    sel = torch.from_numpy(np.random.RandomState(n_rays).choice(1024, n_rays, replace=False))
    r = rays[sel.to(dev)].contiguous()
    bc = syn["bc"].reshape(-1, 3)[sel].contiguous()
    cond = (syn["aud"], syn["expr"], syn["latent"])
    with torch.no_grad():
        ref = oracle.render_rays(r.cpu(), bc, pc, pf, *cond, n_samples=S, n_importance=Ni, dims=dims, taps=True)
    t, u = torch.linspace(0.0, 1.0, S).to(dev), torch.linspace(0.0, 1.0, Ni).to(dev)
    ff = fold_f(*cond)
    out = idn.ops.render_rays_fwd(r, bc.to(dev), pk_c, fold_c(*cond), pk_f, ff, t, u, Ni, taps=True)
    for k in ("rgb0", "disp0", "acc0"):      # nothing is sampled before the coarse composite
        assert rel_err(out[k], ref[k]) < RGB_TOL, k
    prove_render(idn, f"{n_rays} rays {S}+{Ni}", out, ref, pk_f, ff, r, bc.to(dev), lambda z: oracle_fine_pass(pf, dims, r, bc, *cond, z),
                 FLIP_TOL, keys=("rgb_map", "disp_map", "acc_map"))
    fl, _ = flipped_rows(out["tap_inds"], ref["tap_inds"])
    assert abs_err(out["last_weight"][torch.from_numpy(~fl).to(dev)], ref["last_weight"][torch.from_numpy(~fl)]) < W_TOL


@pytest.mark.parametrize("n_rays,jitter,with_fg", [(1, False, False), (6, False, True), (1031, False, False), (517, True, True)])
def test_fused_ray_kernel_equals_the_unfused_path(idn, dev, n_rays, jitter, with_fg):
    """north_star's fused ray-march kernel (csrc/render_fused.hip: both networks, the march between them and the final
    compositing as ONE launch, a ray's depths / raw outputs / weights / cdf in LDS) against the default kernel sequence on
    the same rays: every output and every tap BIT FOR BIT (it runs the same device functions on LDS rows), ragged ray
    counts (groups of four rays, padding rays computed and dropped), per-ray u and stratified depths (perturb = 1), and the
    torso variant's outputs.  The kernel sequence itself is pinned to the reference by the goldens above."""
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(48, 48, seed=7, dims=dims)
    rays = idn.ops.frame_rays(syn["c2w"], 48, 48, syn["focal"], NEAR, FAR, device=dev)
    sel = torch.from_numpy(np.random.RandomState(n_rays).choice(48 * 48, n_rays, replace=False))
    r = rays[sel.to(dev)].contiguous()
    bc = syn["bc"].reshape(-1, 3)[sel].contiguous().to(dev)
    cond = (syn["aud"], syn["expr"], syn["latent"])
    t = torch.linspace(0.0, 1.0, 64).to(dev)
    g = torch.Generator().manual_seed(n_rays)
    u = torch.rand((n_rays, 128), generator=g).to(dev) if jitter else torch.linspace(0.0, 1.0, 128).to(dev)
    t_rand = torch.rand((n_rays, 64), generator=g).to(dev) if jitter else None
    args = (r, bc, pk_c, fold_c(*cond), pk_f, fold_f(*cond), t, u, 128)
    for taps in (True, False):
        # (the reference leaves white_bkgd and lindisp off; the one-kernel path takes them like the sequence does)
        kw = dict(t_rand=t_rand, with_fg=with_fg, taps=taps, white_bkgd=not taps, lindisp=jitter and not taps)
        seq = idn.ops.render_rays_fwd(*args, fused=False, **kw)
        for how in (True, "split"):   # one launch; two launches (coarse + march | fine + compositing, the fine depths through HBM)
            one = idn.ops.render_rays_fwd(*args, fused=how, **kw)
            assert sorted(seq) == sorted(one)
            for k in seq:
                assert torch.isfinite(seq[k].float()).all(), k
                assert torch.equal(seq[k], one[k]), f"{k}: fused={how!r} != kernel sequence (max diff {(seq[k].double() - one[k].double()).abs().max().item():.3e})"
    # what the fused kernel is not built for is refused, not approximated
    with pytest.raises(idn._lib.IdealNerfError, match="fused march"):
        idn.ops.render_rays_fwd(r, bc, pk_c, fold_c(*cond), pk_f, fold_f(*cond), torch.linspace(0.0, 1.0, 32).to(dev), u, 128, fused=True)


def test_philox_table_matches_the_cpu_restatement(idn, dev):
    """The table the in-kernel draws come from (idealnerf_philox_uniform) against oracle/philox.py -- itself pinned by
    Philox4x32-10's published known-answer vectors -- bit for bit: both tables, ragged widths, a row offset beyond 32 bits."""
    from oracle import philox
    for seed, which, row0, n, cols in [(0, 0, 0, 7, 64), (0x0123456789ABCDEF, 1, 0, 129, 128), (2 ** 64 - 1, 1, 32768, 40, 61),
                                       (12345, 0, (1 << 33) + 5, 16, 5), (99, 1, 0, 1, 1)]:
        got = idn.ops.philox_uniform(seed, which, row0, n, cols, dev)
        want = torch.from_numpy(philox.uniform_table(seed, which, row0, n, cols))
        assert got.shape == want.shape and torch.equal(got.cpu(), want), (seed, which, row0, n, cols)
    assert idn.ops.philox_uniform(1, 0, 0, 0, 64, dev).shape == (0, 64)


@pytest.mark.parametrize("n_rays,ray0", [(5, 0), (1031, 77), (40000, 3)])
def test_in_kernel_draws_equal_the_same_draws_as_tensors(idn, dev, n_rays, ray0):
    """perturb > 0 with the draws made INSIDE the kernels (idn_render_args.rng_mode = 1: stratified offsets in the coarse-depth
    kernel, importance draws in the march) against the same render fed the table as t_rand / u tensors -- written by the CPU
    restatement of the generator: every output and every tap BIT FOR BIT, in all three kernel arrangements, across the
    library's internal 32 768-ray passes (40 000 rays: the second pass must continue at row ray0 + 32 768).  The tensor path
    itself is pinned to the reference under jitter by `test_render_rays_golden_jitter` / `rays64_jitter.npz`."""
    from oracle import philox
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    syn = oracle.synthetic_frame(48, 48, seed=7, dims=dims)
    rays = idn.ops.frame_rays(syn["c2w"], 48, 48, syn["focal"], NEAR, FAR, device=dev)
    sel = torch.from_numpy(np.random.RandomState(n_rays).choice(48 * 48, n_rays, replace=n_rays > 48 * 48))
    r = rays[sel.to(dev)].contiguous()
    bc = syn["bc"].reshape(-1, 3)[sel].contiguous().to(dev)
    cond = (syn["aud"], syn["expr"], syn["latent"])
    t = torch.linspace(0.0, 1.0, 64).to(dev)
    seed = 0x5DEECE66D + n_rays
    t_rand = torch.from_numpy(philox.uniform_table(seed, 0, ray0, n_rays, 64)).to(dev)
    u = torch.from_numpy(philox.uniform_table(seed, 1, ray0, n_rays, 128)).to(dev)
    nets = (pk_c, fold_c(*cond), pk_f, fold_f(*cond), t)
    taps = n_rays < 2000
    for how in (False, True, "split"):
        ten = idn.ops.render_rays_fwd(r, bc, *nets, u, 128, t_rand=t_rand, taps=taps, with_fg=True, fused=how)
        drw = idn.ops.render_rays_fwd(r, bc, *nets, None, 128, taps=taps, with_fg=True, fused=how, draws=(seed, ray0))
        assert sorted(ten) == sorted(drw)
        for k in ten:
            assert torch.isfinite(ten[k].float()).all(), k
            assert torch.equal(ten[k], drw[k]), f"{k} (fused={how!r}): in-kernel draws != the table as tensors"
    if taps:   # the draws did move the samples: against the deterministic render the depths differ
        det = idn.ops.render_rays_fwd(r, bc, *nets, torch.linspace(0.0, 1.0, 128).to(dev), 128, taps=True)
        assert not torch.equal(det["tap_z_coarse"], drw["tap_z_coarse"]) and not torch.equal(det["tap_z_samples"], drw["tap_z_samples"])
    # coarse-only renders draw their offsets the same way; the draws replace BOTH tensors
    c_t = idn.ops.render_rays_fwd(r, bc, pk_c, nets[1], None, None, t, None, 0, t_rand=t_rand)
    c_d = idn.ops.render_rays_fwd(r, bc, pk_c, nets[1], None, None, t, None, 0, draws=(seed, ray0))
    assert all(torch.equal(c_t[k], c_d[k]) for k in c_t)
    with pytest.raises(idn._lib.IdealNerfError, match="replaces BOTH"):
        idn.ops.render_rays_fwd(r, bc, *nets, u, 128, draws=(seed, ray0))


def test_in_kernel_draws_in_frame_mode_do_not_depend_on_the_partition(idn, dev):
    """Full-frame mode with in-kernel draws: a ray's row of the draw table is its pixel index, so a frame rendered as one band,
    as two uneven bands (what two ranks would do) or from materialised ray records with the table as tensors is the same frame
    bit for bit; and the module path (`net.in_kernel_draws = True`, the reference's default perturb = 1 in eval) renders
    exactly that frame for the seed it drew from torch's CPU generator."""
    from oracle import philox
    dims, pc, pf, (pk_c, fold_c), (pk_f, fold_f) = _nets(idn, dev)
    H = W = 40
    syn = oracle.synthetic_frame(H, W, seed=3, dims=dims)
    cond = (syn["aud"], syn["expr"], syn["latent"])
    t = torch.linspace(0.0, 1.0, 64).to(dev)
    nets = (pk_c, fold_c(*cond), pk_f, fold_f(*cond), t)
    bc = syn["bc"].reshape(H, W, 3).to(dev)
    seed = 424242
    whole = idn.ops.render_rays_fwd(None, bc.reshape(-1, 3), *nets, None, 128, draws=(seed, 0),
                                    frame=idn.ops.make_frame(syn["c2w"], H, W, syn["focal"], NEAR, FAR))
    parts = []
    for row0, nrows in ((0, 13), (13, 27)):
        f = idn.ops.make_frame(syn["c2w"], H, W, syn["focal"], NEAR, FAR, row0, nrows)
        parts.append(idn.ops.render_rays_fwd(None, bc[row0:row0 + nrows].reshape(-1, 3).contiguous(), *nets, None, 128, draws=(seed, row0 * W), frame=f))
    rays = idn.ops.frame_rays(syn["c2w"], H, W, syn["focal"], NEAR, FAR, device=dev)
    ten = idn.ops.render_rays_fwd(rays, bc.reshape(-1, 3), *nets, torch.from_numpy(philox.uniform_table(seed, 1, 0, H * W, 128)).to(dev), 128,
                                  t_rand=torch.from_numpy(philox.uniform_table(seed, 0, 0, H * W, 64)).to(dev))
    for k in whole:
        assert torch.equal(whole[k], torch.cat([p[k] for p in parts], 0)), k
        assert torch.equal(whole[k], ten[k]), k
    # the module path: render_dynamic_face in eval mode with the reference's default flags (perturb = 1, helper.py:70)
    from idealnerf_amd.audio_exp_nerf import Network
    from idealnerf_amd.helper import RenderConfig
    net = Network(H, W, syn["focal"], NEAR, FAR, 512, None, 64, 128, args=RenderConfig(perturb=1.0, chunk=512, near=NEAR, far=FAR)).to(dev)
    net.face_nerf_coarse.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(2, dims)))
    net.face_nerf_fine.load_state_dict(scale_sigma(oracle.xavier_facenerf_params(3, dims)))
    net.eval()
    assert net.in_kernel_draws is False        # opt-in
    net.in_kernel_draws = True
    call = lambda: net.render_dynamic_face(H, W, syn["focal"], expr=syn["expr"].to(dev), poses=syn["c2w"], latent_code=syn["latent"].to(dev),
                                           render_poses=syn["c2w"][:3, :4], chunk=512, near=NEAR, far=FAR, bc_rgb=bc, aud_para=syn["aud"].to(dev))
    torch.manual_seed(5)
    seed_net = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.manual_seed(5)
    with torch.no_grad():
        rgb = call()[0]
        again = call()[0]          # the next call draws the next seed: another frame of the same distribution
    cd = tuple(v.to(dev) for v in cond)
    c, f = net.face_nerf_coarse, net.face_nerf_fine
    want = idn.ops.render_rays_fwd(None, bc.reshape(-1, 3), c.packed_weights(), c.folded_bias(cd[0], cd[1], cd[2]), f.packed_weights(),
                                   f.folded_bias(cd[0], cd[1], cd[2]), t, None, 128, precision=c.prec_code, draws=(seed_net, 0),
                                   frame=idn.ops.make_frame(syn["c2w"], H, W, syn["focal"], NEAR, FAR))
    assert torch.equal(rgb.reshape(-1, 3), want["rgb_map"])
    assert not torch.equal(again, rgb) and float((again - rgb).abs().mean()) < 0.05


def test_train_forward_defines_every_row_of_the_activation_slab(idn, dev):
    """The weight-gradient GEMMs contract over all p_pad rows of the saved activations, so the padding
    rows of a ragged pass must hold finite numbers (their deltas are zero, but 0 x NaN is not): fill
    the slab with NaN, run the training forward on 37 x 5 points, and nothing non-finite may remain."""
    import ctypes as C
    lib = idn._lib.load()
    dims = oracle.facenerf_dims()
    net = idn.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76)
    net.load_state_dict(oracle.xavier_facenerf_params(7, dims))
    net = net.to(dev)
    syn = oracle.synthetic_frame(32, 32, seed=1, dims=dims)
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)[:37].contiguous()
    n, S = 37, 5
    z = idn.ops.coarse_depths(rays, torch.linspace(0.0, 1.0, S).to(dev))
    folded = net.folded_bias(syn["aud"].to(dev), syn["expr"].to(dev), syn["latent"].to(dev))
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=dev)
    # both activation-saving kernels (the six-piece bf16 one first, the fp32 one last: its slab is examined below)
    for prec_name, code in (("bf16x6", BF16X6_CODE), ("f32", 0)):
        acts = torch.full((lib.idealnerf_train_acts_floats(n * S),), float("nan"), dtype=torch.float32, device=dev)
        rc = lib.idealnerf_query_rays_train_fwd(net.packed_weights(prec_name).data_ptr(), folded.data_ptr(), code, rays.data_ptr(),
                                                z.data_ptr(), n, S, raw.data_ptr(), acts.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.idealnerf_last_error()
        assert bool(torch.isfinite(acts[:256 * 2560]).all()) and bool(torch.isfinite(raw).all()), prec_name
        if prec_name == "bf16x6":
            acts6 = acts
    assert rel_err(acts6[:256 * 2560], acts[:256 * 2560]) < 1e-5       # the two kernels save the same activations
    # ... and the same ReLU masks (a8-wide layers: four dwords per lane, the 128-wide ones: two; a pre-activation within
    # rounding of zero may fall on either side in the two arithmetics)
    m6, m32 = (a[256 * 2560:].view(torch.int32).reshape(11, 8, 64, 4) for a in (acts6, acts))
    diff = (m6 ^ m32)
    diff[8:, :, :, 2:] = 0
    bits = sum(int(((diff >> b) & 1).sum()) for b in range(32))
    assert bits <= 4, bits
    assert acts.numel() == 256 * (2560 + 88)   # 185 points -> p_pad = 256 rows of 2560 columns + 88 floats of ReLU mask bits
    assert bool(torch.isfinite(acts[:256 * 2560]).all()) and bool(torch.isfinite(raw).all())   # the tail is bit masks, not floats
    # the packed ReLU masks say exactly which saved activations are positive: layer a3 (id 2), every point and channel
    p_pad = 256
    a3 = acts[(2 * 64 + 2 * 256) * p_pad:(2 * 64 + 3 * 256) * p_pad].reshape(p_pad, 256)            # x0, dir, a1, a2, then a3
    words = acts[2560 * p_pad:].view(torch.int32).reshape(11, p_pad // 32, 64, 4)[2]                  # [wave tile, lane, dword]
    pt = torch.arange(p_pad, device=dev)
    for T_ in (0, 3, 7):
        for r in (0, 5, 15):
            for h_ in (0, 1):
                ch = 32 * T_ + (r & 3) + 8 * (r >> 2) + 4 * h_
                w = words[pt // 32, (pt % 32) + 32 * h_, T_ >> 1]
                off = (w >> (31 - (16 * (T_ & 1) + r))) & 1
                assert torch.equal(off == 0, a3[:, ch] > 0), (T_, r, h_)


def test_head_torso_composite_bf16_modes_psnr(idn, dev):
    """BASELINE config 5 end to end: the head + torso composite with the bf16 MLP modes against the same
    composite in exact fp32, judged by PSNR.  On this sharp scene the importance sampling amplifies
    bf16x3's 1.5e-5 on the coarse raw output: a few per cent of the rays move by 1e-3 in the fine pass
    (a 1e-5 change of a cdf with 1e-5-wide bins relocates samples), so bf16x3 is characterised here by
    PSNR and its tail, not by the max-norm budget it meets on the reference's golden frame."""
    net, syn, P, dims, d = _torso_setup(idn, dev, n=256)
    x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
         d["expr"][None], d["latent"], torch.tensor([1]))
    outs = {}
    nets = (net.face_nerf_coarse, net.face_nerf_fine, net.torso_coarse_nerf, net.torso_fine_nerf)
    net.train()   # the ray-batch branch of forward (render_poses=None); autograd is off, so the inference kernels run
    with torch.no_grad():
        for prec in ("f32", "bf16x3", "bf16"):
            for m in nets:
                m.precision = prec
            outs[prec] = [o.cpu().numpy().astype(np.float64) for o in net([x, 0, 4])]
    psnr = lambda a, b: -10.0 * np.log10(max(((a - b) ** 2).mean(), 1e-30))
    p3, p1 = psnr(outs["bf16x3"][0], outs["f32"][0]), psnr(outs["bf16"][0], outs["f32"][0])
    e3 = np.abs(outs["bf16x3"][0] - outs["f32"][0]).max(1)
    print(f"\nhead+torso composite vs fp32: bf16x3 PSNR {p3:.1f} dB (max {e3.max():.1e}, rays > 1e-4: {(e3 > 1e-4).mean():.1%}), "
          f"plain bf16 PSNR {p1:.1f} dB; coarse composite bf16x3 max {np.abs(outs['bf16x3'][1] - outs['f32'][1]).max():.1e}")
    assert p3 > 60.0 and p1 > 40.0
    assert rel_err(outs["bf16x3"][1], outs["f32"][1]) < RGB_TOL   # the coarse composite has no importance sampling before it
    # the same three composites against the CPU oracle (pinned to the reference; torso conditioning pinned by
    # tests/golden/torso_signal.npz), each at its mode's budget
    for m in nets:
        m.precision = "f32"
    _, ref, ref0 = prove_head_torso(idn, "head+torso fp32 (256 rays)", net, P, dims, d, dev, outs["f32"][0], FLIP_TOL_SHARP)
    ref, ref0 = ref.numpy().astype(np.float64), ref0.numpy().astype(np.float64)
    o3, o1 = psnr(outs["bf16x3"][0], ref), psnr(outs["bf16"][0], ref)
    print(f"head+torso composite vs CPU oracle: fp32 max rel {rel_err(outs['f32'][0], ref):.1e}, bf16x3 PSNR {o3:.1f} dB, "
          f"plain bf16 PSNR {o1:.1f} dB")
    assert rel_err(outs["f32"][1], ref0) < RGB_TOL
    assert o3 > 60.0 and o1 > 40.0
    assert rel_err(outs["bf16x3"][1], ref0) < RGB_TOL


def test_mixed6_and_bf16x6_on_the_sharp_scene(idn, dev):
    """On the sharp head+torso scene (where plain bf16x3 leaves the budget): "bf16x6" (every network fp32-grade on the bf16
    pipe) and "mixed6" (bf16x6 coarse network, which drives the sampling, + bf16x3 fine network) against the fp32 kernels
    and against the CPU oracle at the budget the fp32 kernels are held to."""
    net, syn, P, dims, d = _torso_setup(idn, dev, n=512)
    x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
         d["expr"][None], d["latent"], torch.tensor([1]))
    net.train()   # ray-batch branch; autograd is off, so the inference kernels run
    outs = {}
    with torch.no_grad():
        for mode in ("f32", "bf16x6", "mixed6"):
            idn.set_render_precision(net, mode)
            outs[mode] = [o.cpu().numpy().astype(np.float64) for o in net([x, 0, 4])]
    assert net.face_nerf_coarse.precision == "bf16x6" and net.face_nerf_fine.precision == "bf16x3"
    # (the scene is built so that one flipped importance index moves a pixel by ~1e-4: every ray beyond 1e-4 is shown to
    #  carry a flip, and on the oracle's own positions every mode is inside 1e-4 -- fp32 kernels 0.4-0.8 % of 512 rays
    #  flipped, the bf16x6 coarse network about twice that, as on the reference's golden frame: 3e-5 vs 1.5e-5 of the indices)
    stats = {}
    for mode in ("f32", "bf16x6", "mixed6"):
        idn.set_render_precision(net, mode)
        res, ref, ref0 = prove_head_torso(idn, f"head+torso {mode}", net, P, dims, d, dev, outs[mode][0],
                                          FLIP_TOL_SHARP if mode == "f32" else 2 * FLIP_TOL_SHARP)
        stats[mode] = res
        print(f"  {mode:7s}: vs the fp32 kernels {rel_err(outs[mode][0], outs['f32'][0]):.2e}, coarse composite vs oracle {rel_err(outs[mode][1], ref0):.2e}")
        assert res["beyond"] < (0.01 if mode == "f32" else 0.03) * 512, mode
        assert rel_err(outs[mode][1], ref0) < RGB_TOL, mode      # no sampling before the coarse composite
    idn.set_render_precision(net, "f32")
    assert_default_precision_allowed(default_precision_criterion("sharp head + torso scene", stats["f32"], stats["bf16x6"], 512),
                                     "the sharp head + torso scene")


def test_mixed_precision_keeps_the_rgb_budget_on_the_sharp_scene(idn, dev, golden):
    """"mixed" = coarse network in exact fp32 (its output drives the importance sampling, which amplifies
    arithmetic noise), fine network in bf16x3 (3/4 of the samples, nothing is sampled after it).  On the
    scene where plain bf16x3 leaves the 1e-4 budget, mixed stays inside it against fp32; on the reference's
    golden frame it stays inside it against the reference, with the reference's own sample positions."""
    net, syn, P, dims, d = _torso_setup(idn, dev, n=512)
    x = (d["batch_rays"][None], d["batch_rays_torso"][None], d["target"], d["bg"], d["auds"][None], None, d["pose"],
         d["expr"][None], d["latent"], torch.tensor([1]))
    net.train()   # ray-batch branch; autograd is off, so the inference kernels run
    outs = {}
    with torch.no_grad():
        for mode in ("f32", "mixed"):
            idn.set_render_precision(net, mode)
            outs[mode] = [o.cpu().numpy().astype(np.float64) for o in net([x, 0, 4])]
    assert net.face_nerf_coarse.precision == "f32" and net.face_nerf_fine.precision == "bf16x3"
    e = rel_err(outs["mixed"][0], outs["f32"][0])
    print(f"\nmixed vs fp32 on the head+torso scene (512 rays): max rel err {e:.2e}")
    assert e < RGB_TOL
    np.testing.assert_array_equal(outs["mixed"][1], outs["f32"][1])   # the coarse composite is the same arithmetic
    # This scene is built to be sharp (sigma gain 100 on the head): a fine sample relocated by one flipped
    # importance index changes its pixel by ~1e-4, whichever fp32 implementation flipped it -- so the two
    # exact-fp32 evaluations (CPU BLAS vs fp32 MFMA) differ by 2e-4 on the worst of 512 rays, and `mixed`
    # inherits exactly that (it agrees with the fp32 kernel to 4e-6, above).  Proved, not assumed (parity_proof):
    # on the oracle's own sample positions both modes are inside 1e-4 on every ray, and each end-to-end ray beyond
    # 1e-4 has a flipped index in the head or the torso render; the coarse composite (no sampling before it)
    # is inside 1e-4 outright.
    for mode in ("f32", "mixed"):
        idn.set_render_precision(net, mode)
        res, ref, ref0 = prove_head_torso(idn, f"head+torso {mode}", net, P, dims, d, dev, outs[mode][0], FLIP_TOL_SHARP)
        assert res["beyond"] < 0.01 * 512, mode
        assert rel_err(outs[mode][1], ref0) < RGB_TOL
    idn.set_render_precision(net, "f32")

    g = golden("frame32")
    dims32 = oracle.facenerf_dims()
    syn32 = oracle.synthetic_frame(32, 32, seed=0, dims=dims32)
    cond = [t.to(dev) for t in (syn32["aud"], syn32["expr"], syn32["latent"])]
    packs = []
    for seed, prec in ((2, 0), (3, 1)):   # coarse IDN_PREC_F32, fine IDN_PREC_BF16X3
        sd = {k: t.to(dev).contiguous() for k, t in scale_sigma(oracle.xavier_facenerf_params(seed, dims32)).items()}
        ps = idn.ops.params_struct(sd, 64, 76, 32)
        packs.append((idn.ops.pack_weights(ps, dev, prec), idn.ops.fold_conditioning(ps, *cond, dev), sd))
    rays = idn.ops.frame_rays(syn32["c2w"], 32, 32, syn32["focal"], NEAR, FAR, device=dev)
    out = idn.ops.render_rays_fwd(rays, syn32["bc"].reshape(-1, 3).to(dev), packs[0][0], packs[0][1], packs[1][0], packs[1][1],
                                  torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev), 128,
                                  taps=True, precision=0, precision_fine=1)
    assert rel_err(out["rgb_map"], g["rgb"].reshape(-1, 3)) < RGB_TOL and rel_err(out["rgb0"], g["rgb0"].reshape(-1, 3)) < 1e-5
    flips = (out["tap_inds"].cpu().numpy() != g["tap_inds"].astype(np.int64)).mean()
    assert flips < FLIP_TOL   # the fp32 kernel's own flip rate, not bf16x3's


@pytest.mark.parametrize("seed", [11, 14, 16, 21])
def test_random_scenes_fp32_and_mixed_vs_oracle(idn, dev, seed):
    """Fresh weights, pose, conditioning and a random ray subset per seed: the fp32 and the mixed mode
    against the CPU oracle (which is pinned to the reference), every output finite."""
    dims = oracle.facenerf_dims()
    rs = np.random.RandomState(seed)
    # (seeds: round 2's 12 and 13 drew EMPTY volumes -- a Xavier density head is positive or negative over the whole volume --
    #  so the frame equalled the background bit for bit and the comparison was vacuous; 14, 16, 21 hide a quarter of the
    #  background in both passes, 11 is the sharp one: dense coarse pass, nearly empty fine pass)
    pc = scale_sigma(oracle.xavier_facenerf_params(100 + seed, dims), float(rs.uniform(30, 300)), float(rs.uniform(0.1, 0.4)))
    pf = scale_sigma(oracle.xavier_facenerf_params(200 + seed, dims), float(rs.uniform(30, 300)), float(rs.uniform(0.1, 0.4)))
    syn = oracle.synthetic_frame(32, 32, seed=seed, dims=dims)
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    sel = torch.from_numpy(rs.choice(1024, 160, replace=False))
    r = rays[sel.to(dev)].contiguous()
    bc = syn["bc"].reshape(-1, 3)[sel].contiguous()
    cond = (syn["aud"], syn["expr"], syn["latent"])
    with torch.no_grad():
        ref = oracle.render_rays(r.cpu(), bc, pc, pf, *cond, n_samples=64, n_importance=128, dims=dims, taps=True)
    assert max(float((ref["rgb_map"] - bc).abs().mean()), float((ref["rgb0"] - bc).abs().mean())) > 0.02, "the volume must hide part of the background"
    t, u = torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev)
    for mode, (prec_c, prec_f) in (("f32", (0, 0)), ("mixed", (0, 1))):
        packs = []
        for p, prec in ((pc, prec_c), (pf, prec_f)):
            sd = {k: v.to(dev).contiguous() for k, v in p.items()}
            ps = idn.ops.params_struct(sd, 64, 76, 32)
            packs.append((idn.ops.pack_weights(ps, dev, prec), idn.ops.fold_conditioning(ps, *(c.to(dev) for c in cond), dev), sd))
        out = idn.ops.render_rays_fwd(r, bc.to(dev), packs[0][0], packs[0][1], packs[1][0], packs[1][1], t, u, 128,
                                      precision=prec_c, precision_fine=prec_f, taps=True)
        for k in ("rgb_map", "rgb0", "disp_map", "acc_map", "last_weight", "z_std"):
            assert bool(torch.isfinite(out[k]).all()), (mode, k)
        prove_render(idn, f"scene {seed} {mode}", out, ref, packs[1][0], packs[1][1], r, bc.to(dev),
                     lambda z: oracle_fine_pass(pf, dims, r, bc, *cond, z), FLIP_TOL_SHARP, precision_fine=prec_f)
        assert rel_err(out["rgb0"], ref["rgb0"]) < RGB_TOL, mode   # the coarse composite has no sampling before it


# --------------------------------------------------------------------------- empty space (most rays of a real head frame)
# Where the volume is empty a compositing weight is alpha * T with alpha = 1 - exp(-1e-6 * dz |d|) ~ 1e-8: in fp32 that is 0 or
# one ulp below 1 (6e-8) by the last bit of exp, whichever code evaluates it, so two correct fp32 implementations differ by up to
# 6e-8 per bin THERE.  sample_pdf (helper.py:271-275) adds 1e-5 to each of the 62 interior weights and divides by their sum,
# ~6.2e-4: a cdf edge e moves by up to e * 6e-8 / 6.2e-4 = e * 1e-4, and an importance index flips whenever one of the 128 draws
# (spacing 1/127) lies between the two positions of an edge -- probability 127 * |shift|.  Summed over the 62 edges that is at
# most 127 / 128 * 1e-4 * (1 + 2 + ... + 62) = 0.19 flipped indices per index if EVERY bin of a ray differed in the same
# direction; what the floor implies is therefore a flip-rate bound of 0.2 for empty rays, and the measured rate (a few 1e-3:
# the two exps agree on most arguments) is reported, not bounded more tightly.  None of it can move a pixel: every sample the
# flips relocate sits in empty space, so end to end EVERY ray is within 1e-4 (in fact 1e-6) -- that is what is asserted.
EMPTY_SPACE_FLIP_BOUND = 0.2


def _empty_space_scene(idn, dev, seed, empty_share, gain=200.0):
    """160 rays of a 32 x 32 frame through ONE density field -- the coarse and the fine network carry the same weights, as a
    trained pair agrees on where the head is -- whose density head is shifted so that `empty_share` of the volume has
    sigma <= 0 (1.0: the whole volume; 0.5: positive in about half of it, in the spatially coherent blobs a smooth MLP draws:
    every ray crosses empty stretches, whose bins sit on sample_pdf's floor, next to stretches that carry mass).
    (With two INDEPENDENT random fields the fine network is dense where the coarse one reports nothing, and the reference's own
    formula then moves a pixel by up to 2e-3 for a last-bit change of a coarse weight -- reproduced on the CPU oracle alone by
    perturbing its coarse weights by 1e-7: a property of such a scene, not of an implementation.)"""
    dims = oracle.facenerf_dims()
    rs = np.random.RandomState(seed)
    syn = oracle.synthetic_frame(32, 32, seed=seed, dims=dims)
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    sel = torch.from_numpy(rs.choice(1024, 160, replace=False))
    r = rays[sel.to(dev)].contiguous()
    bc = syn["bc"].reshape(-1, 3)[sel].contiguous()
    cond = (syn["aud"], syn["expr"], syn["latent"])
    rc = r.cpu()
    z = oracle.coarse_depths(rc[:, 6:7], rc[:, 7:8], 64, None)
    pts = rc[:, None, 0:3] + rc[:, None, 3:6] * z[:, :, None]
    p = scale_sigma(oracle.xavier_facenerf_params(300 + seed, dims), gain, 0.0)
    with torch.no_grad():
        sig = oracle.render_oracle._query(p, pts, rc[:, -3:], *cond, dims)[..., 3].reshape(-1)
    if empty_share >= 1.0:
        # (the fine pass evaluates the field between the coarse samples: leave half the field's range as margin)
        p["alpha_linear.bias"] = torch.full_like(p["alpha_linear.bias"], -float(sig.max()) - 0.5 * float(sig.max() - sig.min()))
    else:
        p["alpha_linear.bias"] = torch.full_like(p["alpha_linear.bias"], -float(torch.quantile(sig, empty_share)))
    pc = pf = p
    with torch.no_grad():
        ref = oracle.render_rays(rc, bc, pc, pf, *cond, n_samples=64, n_importance=128, dims=dims, taps=True)
    sd = {k: v.to(dev).contiguous() for k, v in p.items()}
    ps = idn.ops.params_struct(sd, 64, 76, 32)
    pk, fold = idn.ops.pack_weights(ps, dev, 0), idn.ops.fold_conditioning(ps, *(c.to(dev) for c in cond), dev)
    out = idn.ops.render_rays_fwd(r, bc.to(dev), pk, fold, pk, fold,
                                  torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev), 128, taps=True)
    res = prove_render(idn, f"empty share {empty_share:.1f}, scene {seed}", out, ref, pk, fold, r, bc.to(dev),
                       lambda zf: oracle_fine_pass(pf, dims, r, bc, *cond, zf), EMPTY_SPACE_FLIP_BOUND)
    return out, ref, bc, res["rgb_map"]


@pytest.mark.parametrize("seed", [12, 13])
def test_empty_volume_is_the_background_on_every_ray(idn, dev, seed):
    """sigma <= 0 everywhere (helper.py:271-275 runs on weights of ~1e-8 over its 1e-5 floor): parity_proof's three legs on
    every ray, the frame equal to the background to 1e-6 in both passes, EVERY ray within 1e-4 of the oracle end to end --
    and the importance-index flip rate reported against the bound the floor implies (derivation above)."""
    out, ref, bc, res = _empty_space_scene(idn, dev, seed, 1.0)
    assert float((ref["rgb_map"] - bc).abs().max()) < 1e-6, "the oracle's volume is not empty"
    for k in ("rgb_map", "rgb0"):
        assert abs_err(out[k], bc) < 1e-6, k
    assert res["beyond"] == 0 and res["e2e_max"] < 1e-5, res
    assert abs_err(out["acc_map"], ref["acc_map"]) < 1e-6
    print(f"  empty volume {seed}: interior flip rate {res['flip_rate']:.2e} (bound from the 1e-5 floor: {EMPTY_SPACE_FLIP_BOUND})")


@pytest.mark.parametrize("seed", [12, 13])
def test_half_empty_volume_vs_oracle(idn, dev, seed):
    """The density is positive in about half of the volume: every ray has empty stretches whose bins sit on the 1e-5 floor
    next to bins that carry mass.  The three legs on every ray, every ray within 1e-4 end to end (the rays that cross
    nothing are the background to 1e-6), the flip rate reported against the floor's bound."""
    out, ref, bc, res = _empty_space_scene(idn, dev, seed, 0.5, gain=40.0)
    vis = (ref["rgb_map"] - bc).abs().max(1)[0]
    assert float(vis.mean()) > 0.02, "half of the volume must carry density"
    floor_bins = float((ref["tap_weights_coarse"][:, 1:-1] < 1e-6).float().mean())
    assert 0.2 < floor_bins < 0.9, f"{floor_bins:.0%} of the coarse bins sit on sample_pdf's floor: not a half-empty scene"
    empty_rays = (ref["tap_weights_fine"][:, :-1].sum(1) < 1e-6)
    if bool(empty_rays.any()):
        assert abs_err(out["rgb_map"][empty_rays.to(dev)], bc[empty_rays]) < 1e-6
    assert res["beyond"] == 0, res
    assert rel_err(out["rgb0"], ref["rgb0"]) < RGB_TOL
    print(f"  half-empty volume {seed}: {floor_bins:.0%} of the coarse bins on the floor, {int(empty_rays.sum())} of 160 rays cross nothing; interior flip rate {res['flip_rate']:.2e} "
          f"(bound from the 1e-5 floor: {EMPTY_SPACE_FLIP_BOUND})")


# --------------------------------------------------------------------------- fp16x3 arithmetic mode
FP16X3 = 3  # IDN_PREC_FP16X3


def test_fp16x3_facenerf_golden_and_ragged(idn, dev, golden):
    """Three fp16 MFMAs per product (11+11 significand bits per operand): fp32-like error on the network
    output at the bf16x3 speed, for activations inside fp16's range."""
    g = golden("facenerf")
    for name, v in (("c235", dict(dim_aud=64, dim_expr=76, dim_latent=32)), ("c169", dict(dim_aud=106, dim_expr=0, dim_latent=0))):
        dims = oracle.facenerf_dims(**v)
        sd = {k: t.to(dev).contiguous() for k, t in oracle.xavier_facenerf_params(11, dims).items()}
        ps = idn.ops.params_struct(sd, dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])
        opt = lambda k: T(g[k]).to(dev) if k in g else None
        folded = idn.ops.fold_conditioning(ps, T(g[name + "_aud"]).to(dev), opt(name + "_expr"), opt(name + "_latent"), dev)
        out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, FP16X3), folded, T(g[name + "_x"]).to(dev), FP16X3)
        assert rel_err(out, g[name + "_out"]) < 5e-6, name
    dims = oracle.facenerf_dims()
    params = scale_sigma(oracle.xavier_facenerf_params(5, dims), 30.0, 0.1)
    for n in (1, 130, 4099):
        rs = np.random.RandomState(n)
        x = T(rs.uniform(-1, 1, size=(n, 90)).astype(np.float32))
        aud, expr, lat = (T(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
        with torch.no_grad():
            ref = oracle.facenerf_forward(params, x, aud, expr, lat, dims)
        sd = {k: t.to(dev).contiguous() for k, t in params.items()}
        ps = idn.ops.params_struct(sd, 64, 76, 32)
        out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, FP16X3),
                                   idn.ops.fold_conditioning(ps, aud.to(dev), expr.to(dev), lat.to(dev), dev), x.to(dev), FP16X3)
        assert rel_err(out, ref) < 1e-5, n


def test_fp16x3_render_frame32_golden(idn, dev, golden):
    g = golden("frame32")
    dims = oracle.facenerf_dims()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    cond = [t.to(dev) for t in (syn["aud"], syn["expr"], syn["latent"])]
    packs = []
    for seed in (2, 3):
        sd = {k: t.to(dev).contiguous() for k, t in scale_sigma(oracle.xavier_facenerf_params(seed, dims)).items()}
        ps = idn.ops.params_struct(sd, 64, 76, 32)
        packs.append((idn.ops.pack_weights(ps, dev, FP16X3), idn.ops.fold_conditioning(ps, *cond, dev), sd))
    rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
    out = idn.ops.render_rays_fwd(rays, syn["bc"].reshape(-1, 3).to(dev), packs[0][0], packs[0][1], packs[1][0], packs[1][1],
                                  torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev), 128,
                                  taps=True, precision=FP16X3)
    e = rel_err(out["rgb_map"], g["rgb"].reshape(-1, 3))
    flips = (out["tap_inds"].cpu().numpy() != g["tap_inds"].astype(np.int64)).mean()
    print(f"\nfp16x3 frame32: rgb err {e:.2e}, index flip rate {flips:.2e}")
    assert e < RGB_TOL and rel_err(out["rgb0"], g["rgb0"].reshape(-1, 3)) < 1e-5
    assert flips < 3e-4   # measured 6.1e-5 (8 of 131 072)


def test_fp16x3_saturates_finitely_outside_fp16_range(idn, dev):
    """fp16x3's documented limit: activations beyond fp16's range (6.5e4) make its hi/lo split saturate.  The
    result is then wrong but finite (v_cvt_pkrtz never produces inf), and the fp32 and bf16x3 modes of the
    same weights are unaffected -- the failure is loud in a comparison, never a NaN frame."""
    dims = oracle.facenerf_dims()
    params = oracle.xavier_facenerf_params(9, dims)
    params["pts_linears.0.weight"] = params["pts_linears.0.weight"] * 3e5    # first hidden layer ~1e5
    rs = np.random.RandomState(0)
    x = T(rs.uniform(-1, 1, size=(256, 90)).astype(np.float32))
    aud, expr, lat = (T(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
    with torch.no_grad():
        ref = oracle.facenerf_forward(params, x, aud, expr, lat, dims)
    sd = {k: t.to(dev).contiguous() for k, t in params.items()}
    ps = idn.ops.params_struct(sd, 64, 76, 32)
    folded = idn.ops.fold_conditioning(ps, aud.to(dev), expr.to(dev), lat.to(dev), dev)
    outs = {name: idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, code), folded, x.to(dev), code)
            for name, code in (("f32", 0), ("bf16x3", 1), ("fp16x3", 3))}
    assert rel_err(outs["f32"], ref) < 1e-5 and rel_err(outs["bf16x3"], ref) < 1e-4
    assert bool(torch.isfinite(outs["fp16x3"]).all())
    assert rel_err(outs["fp16x3"], ref) > 1e-3     # outside its domain, and visibly so


# --------------------------------------------------------------------------- six-piece bf16 arithmetic mode
BF16X6 = 4  # IDN_PREC_BF16X6


def test_bf16x6_facenerf_golden_ragged_and_range(idn, dev, golden):
    """Six bf16 piece products per fp32 product (weights and activations as the exact sum of three bf16 pieces):
    the fp32 kernel's error on the reference's golden vectors, on ragged point counts, and -- unlike fp16x3 -- on
    activations far outside fp16's range (bf16 pieces have fp32's exponent range)."""
    g = golden("facenerf")
    for name, v in (("c235", dict(dim_aud=64, dim_expr=76, dim_latent=32)), ("c169", dict(dim_aud=106, dim_expr=0, dim_latent=0))):
        dims = oracle.facenerf_dims(**v)
        sd = {k: t.to(dev).contiguous() for k, t in oracle.xavier_facenerf_params(11, dims).items()}
        ps = idn.ops.params_struct(sd, dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])
        opt = lambda k: T(g[k]).to(dev) if k in g else None
        folded = idn.ops.fold_conditioning(ps, T(g[name + "_aud"]).to(dev), opt(name + "_expr"), opt(name + "_latent"), dev)
        x = T(g[name + "_x"]).to(dev)
        out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, BF16X6), folded, x, BF16X6)
        out32 = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, 0), folded, x, 0)
        e6, e32 = rel_err(out, g[name + "_out"]), rel_err(out32, g[name + "_out"])
        print(f"\nbf16x6 FaceNeRF {name}: max rel err vs reference = {e6:.2e} (the fp32 kernel: {e32:.2e})")
        assert e6 < 5e-6, name
    dims = oracle.facenerf_dims()
    params = scale_sigma(oracle.xavier_facenerf_params(5, dims), 30.0, 0.1)
    for n in (1, 33, 130, 4099):
        rs = np.random.RandomState(n)
        x = T(rs.uniform(-1, 1, size=(n, 90)).astype(np.float32))
        aud, expr, lat = (T(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
        with torch.no_grad():
            ref = oracle.facenerf_forward(params, x, aud, expr, lat, dims)
        sd = {k: t.to(dev).contiguous() for k, t in params.items()}
        ps = idn.ops.params_struct(sd, 64, 76, 32)
        out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, BF16X6),
                                   idn.ops.fold_conditioning(ps, aud.to(dev), expr.to(dev), lat.to(dev), dev), x.to(dev), BF16X6)
        assert out.shape == (n, 4) and rel_err(out, ref) < 1e-5, n
    # first hidden layer ~1e5: fp16x3 saturates here (test_fp16x3_saturates_finitely_outside_fp16_range); this mode does not care
    params = oracle.xavier_facenerf_params(9, dims)
    params["pts_linears.0.weight"] = params["pts_linears.0.weight"] * 3e5
    rs = np.random.RandomState(0)
    x = T(rs.uniform(-1, 1, size=(256, 90)).astype(np.float32))
    aud, expr, lat = (T(rs.standard_normal(k).astype(np.float32)) for k in (64, 76, 32))
    with torch.no_grad():
        ref = oracle.facenerf_forward(params, x, aud, expr, lat, dims)
    sd = {k: t.to(dev).contiguous() for k, t in params.items()}
    ps = idn.ops.params_struct(sd, 64, 76, 32)
    folded = idn.ops.fold_conditioning(ps, aud.to(dev), expr.to(dev), lat.to(dev), dev)
    out = idn.ops.facenerf_fwd(idn.ops.pack_weights(ps, dev, BF16X6), folded, x.to(dev), BF16X6)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("dims_kw,seed,n_rays,S", [(dict(dim_aud=64, dim_expr=79, dim_latent=32), 22, 512, 192),
                                                    (dict(dim_aud=106, dim_expr=0, dim_latent=0), 24, 517, 192),
                                                    (dict(dim_aud=64, dim_expr=76, dim_latent=32), 3, 4096, 64)])
def test_bf16x6_launches_are_bit_identical_and_agree_with_fp32(idn, dev, dims_kw, seed, n_rays, S):
    """The six-piece kernels make their bf16 pieces with inline-asm conversions whose distance from the MFMA that reads
    them (two wait states: tools/valu_mfma_hazard_ubench.hip) is a property of the code's structure, audited on the compiled
    ISA in the CPU suite -- which a GPU box without hipcc cannot run.  This is the same guard on the device: a violated
    hazard showed as results that differ FROM RUN TO RUN and by orders of magnitude (round 3: 1e29).  Eight launches of the
    bf16x6 forward on every CU must be bit-identical, and fp32-grade against the fp32 MFMA kernel on the same inputs."""
    dims = oracle.facenerf_dims(**dims_kw)
    p = oracle.xavier_facenerf_params(seed, dims)
    p["alpha_linear.weight"] = p["alpha_linear.weight"] * 100.0
    sd = {k: v.to(dev).contiguous() for k, v in p.items()}
    ps = idn.ops.params_struct(sd, dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])
    rs = np.random.RandomState(1)
    cond = [None if not d else T(rs.standard_normal(d).astype(np.float32)).to(dev) for d in (dims["dim_aud"], dims["dim_expr"], dims["dim_latent"])]
    folded = idn.ops.fold_conditioning(ps, *cond, dev)
    syn = oracle.synthetic_frame(64, 64, seed=4)
    rays = idn.ops.frame_rays(syn["c2w"], 64, 64, syn["focal"], NEAR, FAR, device=dev)[:n_rays].contiguous()
    z = idn.ops.coarse_depths(rays, torch.linspace(0, 1, S).to(dev))
    pk6, pk32 = idn.ops.pack_weights(ps, dev, BF16X6_CODE), idn.ops.pack_weights(ps, dev, 0)
    ref = idn.ops.query_rays_fwd(pk32, folded, rays, z, 0)
    outs = [idn.ops.query_rays_fwd(pk6, folded, rays, z, BF16X6_CODE).clone() for _ in range(8)]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[0]).all())
    for i, o in enumerate(outs[1:], 1):
        assert torch.equal(o, outs[0]), f"launch {i} differs from launch 0"
    e = rel_err(outs[0], ref)
    print(f"\n  bf16x6 vs fp32 kernel, {n_rays} rays x {S}: {e:.2e}")
    assert e < 1e-5


def test_bf16x6_render_frame32_golden(idn, dev, golden):
    """Whole render path in the six-piece mode against the reference's golden frame: RGB inside the budget, the
    importance-sample indices flipping at the fp32 kernel's rate."""
    g = golden("frame32")
    dims = oracle.facenerf_dims()
    syn = oracle.synthetic_frame(32, 32, seed=0, dims=dims)
    cond = [t.to(dev) for t in (syn["aud"], syn["expr"], syn["latent"])]
    res = {}
    for code in (BF16X6, 0):
        packs = []
        for seed in (2, 3):
            sd = {k: t.to(dev).contiguous() for k, t in scale_sigma(oracle.xavier_facenerf_params(seed, dims)).items()}
            ps = idn.ops.params_struct(sd, 64, 76, 32)
            packs.append((idn.ops.pack_weights(ps, dev, code), idn.ops.fold_conditioning(ps, *cond, dev)))
        rays = idn.ops.frame_rays(syn["c2w"], 32, 32, syn["focal"], NEAR, FAR, device=dev)
        out = idn.ops.render_rays_fwd(rays, syn["bc"].reshape(-1, 3).to(dev), packs[0][0], packs[0][1], packs[1][0], packs[1][1],
                                      torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev), 128,
                                      taps=True, precision=code)
        res[code] = (rel_err(out["rgb_map"], g["rgb"].reshape(-1, 3)), rel_err(out["rgb0"], g["rgb0"].reshape(-1, 3)),
                     float((out["tap_inds"].cpu().numpy() != g["tap_inds"].astype(np.int64)).mean()))
    print(f"\nbf16x6 frame32: rgb err {res[BF16X6][0]:.2e}, rgb0 {res[BF16X6][1]:.2e}, index flip rate {res[BF16X6][2]:.2e} "
          f"(the fp32 kernel: {res[0][0]:.2e}, {res[0][1]:.2e}, {res[0][2]:.2e})")
    assert res[BF16X6][0] < RGB_TOL and res[BF16X6][1] < 1e-5
    assert res[BF16X6][2] < FLIP_TOL
