"""One-off validation (minutes of CPU time): the WHOLE 512x512 bench frame from the CPU oracle, row band by
row band, against the HIP path in its fp32, bf16x6, fp16x3, mixed and bf16x3 modes.  Writes a JSON summary.

    python tools/full_frame_parity.py [out.json]          (on the GPU box; prints a line per band)
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import idealnerf_amd
from idealnerf_amd import ops, synthetic
from idealnerf_amd.helper import linspace01
import oracle   # the checker

def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "full_frame_parity.json")
    dev = torch.device("cuda")
    H = W = 512
    S, Ni = 64, 128
    syn = synthetic.frame(H, W, seed=0)
    coarse = synthetic.xavier_state_dict(idealnerf_amd.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76), 2, 300.0, 0.3).to(dev)
    fine = synthetic.xavier_state_dict(idealnerf_amd.FaceNeRF(dim_aud=64, dim_latent=32, dim_expr=76), 3, 300.0, 0.3).to(dev)
    aud, expr, latent = (syn[k].to(dev) for k in ("aud", "expr", "latent"))
    t_vals, u = linspace01(S, dev), linspace01(Ni, dev)
    bc = syn["bc"].reshape(-1, 3).contiguous().to(dev)
    frames = {}
    with torch.no_grad():
        for mode, (pc_, pf_) in (("f32", ("f32", "f32")), ("bf16x6", ("bf16x6", "bf16x6")), ("fp16x3", ("fp16x3", "fp16x3")), ("mixed", ("f32", "bf16x3")), ("bf16x3", ("bf16x3", "bf16x3"))):
            coarse.precision, fine.precision = pc_, pf_
            rays = ops.frame_rays(syn["c2w"], H, W, syn["focal"], syn["near"], syn["far"], device=dev)
            o = ops.render_rays_fwd(rays, bc, coarse.packed_weights(), coarse.folded_bias(aud, expr, latent), fine.packed_weights(),
                                    fine.folded_bias(aud, expr, latent), t_vals, u, Ni, precision=coarse.prec_code,
                                    precision_fine=fine.prec_code)
            frames[mode] = {k: o[k].cpu().double().numpy() for k in ("rgb_map", "rgb0")}
    pc = {k: v.detach().cpu() for k, v in coarse.state_dict().items()}
    pf = {k: v.detach().cpu() for k, v in fine.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    band = 32
    ref_rgb, ref_rgb0 = [], []
    t0 = time.time()
    with torch.no_grad():
        for r0 in range(0, H, band):
            ref = oracle.render_frame(H, W, syn["focal"], syn["c2w"], syn["near"], syn["far"], syn["bc"], pc, pf,
                                      syn["aud"], syn["expr"], syn["latent"], rows=(r0, r0 + band), chunk=1024)
            ref_rgb.append(ref["rgb_map"].reshape(-1, 3).double().numpy())
            ref_rgb0.append(ref["rgb0"].reshape(-1, 3).double().numpy())
            print(f"rows {r0}..{r0 + band} done, {time.time() - t0:.0f} s", flush=True)
    ref_rgb, ref_rgb0 = np.concatenate(ref_rgb), np.concatenate(ref_rgb0)
    res = {"frame": "512x512, 64+128 samples, synthetic bench scene", "oracle_seconds": time.time() - t0, "modes": {}}
    for mode, fr in frames.items():
        d = np.abs(fr["rgb_map"] - ref_rgb)
        d0 = np.abs(fr["rgb0"] - ref_rgb0)
        ray = d.max(1)
        res["modes"][mode] = {
            "rgb_max_abs": float(d.max()), "rgb_max_rel": float(d.max() / np.abs(ref_rgb).max()),
            "rgb_psnr_db": float(-10 * np.log10(max((d ** 2).mean(), 1e-30))),
            "rays_beyond_1e-4": int((ray > 1e-4).sum()), "rays": int(len(ray)),
            "rgb0_max_abs": float(d0.max())}
        print(mode, res["modes"][mode], flush=True)
    json.dump(res, open(out_path, "w"), indent=1)

if __name__ == "__main__":
    main()
