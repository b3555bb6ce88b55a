/*
 * idealnerf.h -- C ABI of libidealnerf.so, the MI355X (gfx950) implementation of
 * IDEAL-NeRF's per-ray hot path.
 *
 * The reference (GaryGky/IDEAL-NeRF) is pure Python and has no FFI of its own; its
 * seam for this path is the set of Python callables listed beside each entry point
 * below (paths relative to the reference root).  The Python host layer in
 * ideal-nerf_amd/ mirrors those callables and binds this header through ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major memory owned by the
 *     caller (fp32 unless stated; indices int64), except where marked [host];
 *   - `stream` is a hipStream_t passed as void*; calls enqueue work and return, they
 *     never synchronise, allocate or free;
 *   - return 0 on success, a negative IDN_E* code otherwise; idealnerf_last_error()
 *     returns a thread-local message for the last failure on the calling thread;
 *   - no global mutable state: calls on different streams/devices are independent
 *     (the reference's nn.DataParallel replicas call forward concurrently,
 *     NeRFs/HeadNeRF/test/eval_aud_exp_nerf.py:475).
 *
 * Network architecture is the reference's fixed one: D=8, W=256, skips=[4],
 * multires=10 (63 ch), multires_views=4 (27 ch), use_viewdirs=True
 * (NeRFs/HeadNeRF/train/audio_exp_nerf.py:213-224).  The conditioning widths
 * (dim_aud, dim_expr, dim_latent) are free: they only enter the per-frame folded
 * biases.  Other architectures are rejected with IDN_EUNSUPPORTED.
 */
#ifndef IDEALNERF_H
#define IDEALNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDN_OK 0
#define IDN_EINVAL (-1)       /* bad argument (null pointer, bad size) */
#define IDN_EUNSUPPORTED (-2) /* architecture / size outside the compiled path */
#define IDN_EHIP (-3)         /* a HIP runtime call failed (message has the code) */
#define IDN_EWORKSPACE (-4)   /* workspace too small */

/* Fixed architecture constants (models/face_nerf.py:9-37 at the reference's flags). */
#define IDN_W 256
#define IDN_D 8
#define IDN_PTS_CH 63
#define IDN_VIEWS_CH 27
#define IDN_RAY_FLOATS 11 /* o(3) d(3) near far viewdir(3): audio_exp_nerf.py:412-427 */

/* Arithmetic of the MLP contraction. */
#define IDN_PREC_F32 0    /* v_mfma_f32_32x32x2_f32: exact fp32 fma chains */
#define IDN_PREC_BF16X3 1 /* 3 bf16 MFMAs per product (hi*hi + hi*lo + lo*hi), fp32 accumulate */
#define IDN_PREC_BF16 2   /* plain bf16 MFMA, fp32 accumulate (BASELINE config 5 only) */
#define IDN_PREC_FP16X3 3 /* 3 fp16 MFMAs per product (11+11 significand bits per operand): ~5e-7 on the
                             network output at the bf16x3 speed; activations must stay below 6.5e4 */
#define IDN_PREC_BF16X6 4 /* 6 bf16 MFMAs per product: weights and activations as the exact sum of three bf16 pieces
                             (24 significand bits, fp32's range), fp32 accumulate: fp32-grade (error <= 2^-23 per product) */

int idealnerf_version(void);
const char* idealnerf_last_error(void);

/* Sizes (in floats) of the two per-network device buffers the kernels consume. */
size_t idealnerf_packed_weight_floats(int precision);
size_t idealnerf_folded_bias_floats(void);

/*
 * The 24 parameter tensors of one FaceNeRF in state_dict order, nn.Linear layout
 * [out, in] (models/face_nerf.py:27-37):
 *   pts_w[i]/pts_b[i], i=0..7      pts_linears.i.{weight,bias}
 *   views_w[i]/views_b[i], i=0..2  views_linears.i.{weight,bias}
 *   alpha_w/alpha_b                alpha_linear.{weight,bias}
 *   rgb_w/rgb_b                    rgb_linear.{weight,bias}
 * feature_linear.* exists in the state_dict but is never applied (face_nerf.py:34 vs :66)
 * and is therefore not part of this struct.
 */
typedef struct idn_facenerf_params {
    const float* pts_w[8];
    const float* pts_b[8];
    const float* views_w[3];
    const float* views_b[3];
    const float* alpha_w;
    const float* alpha_b;
    const float* rgb_w;
    const float* rgb_b;
    int dim_aud, dim_expr, dim_latent; /* widths of the per-frame conditioning vectors */
} idn_facenerf_params;

/*
 * Re-lay the per-point part of the weights into MFMA-fragment order
 * (`packed`, idealnerf_packed_weight_floats() floats).  Run once per weight update.
 * Replaces nothing in the reference; it is the load-time half of
 * FaceNeRF.forward (models/face_nerf.py:40-80).
 */
int idealnerf_pack_weights(const idn_facenerf_params* p, int precision, float* packed, void* stream);

/*
 * Fold the per-frame conditioning vectors into bias vectors
 * (`folded`, idealnerf_folded_bias_floats() floats):
 *   b0' = b0 + W0[:, 63:]  . [aud | expr/3 | latent]        (face_nerf.py:45-55,58)
 *   b5' = b5 + W5[:, 63:63+C] . [aud | expr/3 | latent]     (face_nerf.py:61)
 *   bv' = bv + Wv0[:, 283:] . expr/3                        (face_nerf.py:68-70)
 * aud / expr / latent may be NULL exactly when their width is 0 (or, for expr and
 * latent, when the caller passes None as the reference allows).  Run once per frame.
 */
int idealnerf_fold_conditioning(const idn_facenerf_params* p, const float* aud, const float* expr,
                                const float* latent, float* folded, void* stream);

/*
 * FaceNeRF.forward (models/face_nerf.py:40-80): x[n, 90] = [gamma10(pts) | gamma4(dir)]
 * -> out[n, 4] = (rgb_raw, sigma_raw).
 */
int idealnerf_facenerf_fwd(const float* packed, const float* folded, int precision, const float* x,
                           int64_t n, float* out, void* stream);

/*
 * Network.run_network with the embedders fused (audio_exp_nerf.py:376-394 +
 * helper.py:174-224): for ray r and sample s, point = o_r + d_r * z[r, s] is
 * encoded in registers and evaluated; raw[r, s, :] = (rgb_raw, sigma_raw).
 */
int idealnerf_query_rays_fwd(const float* packed, const float* folded, int precision, const float* rays,
                             const float* z, int64_t n_rays, int n_samples, float* raw, void* stream);

/* Same, for callers that hold the sample points themselves (the literal signature of
 * Network.run_network, audio_exp_nerf.py:376): pts[n_rays, n_samples, 3],
 * viewdirs[n_rays, 3] (unit vectors, one per ray). */
int idealnerf_query_points_fwd(const float* packed, const float* folded, int precision, const float* pts,
                               const float* viewdirs, int64_t n_rays, int n_samples, float* raw, void* stream);

/* get_rays + ray-record assembly for rows [row0, row0+nrows) of an HxW frame
 * (helper.py:228-243, audio_exp_nerf.py:396-427; default principal point W/2,H/2 when
 * cx/cy < 0).  c2w is 12 floats, row-major [3,4], [host] memory.  rays_out[nrows*W, 11]. */
int idealnerf_frame_rays(const float* c2w, int H, int W, float focal, float cx, float cy, float near_, float far_,
                         int row0, int nrows, float* rays_out, void* stream);

/* Coarse depths (audio_exp_nerf.py:306-330): z[r,s] = near_r (1-t_s) + far_r t_s -- or, with
 * lindisp != 0, linear in inverse depth: 1 / ((1/near_r)(1-t_s) + (1/far_r) t_s) (:309-310) -- with
 * optional stratified jitter from t_rand[n_rays, n_samples] (NULL = perturb 0).
 * t_vals[n_samples] is the caller's torch.linspace(0,1,S) buffer. */
int idealnerf_coarse_depths(const float* rays, const float* t_vals, const float* t_rand, int lindisp, int64_t n_rays,
                            int n_samples, float* z, void* stream);

/*
 * raw2outputs (NeRFs/HeadNeRF/train/baseline.py:325-375; rgb_fg: NeRFs/TorsoNeRF/run_nerf.py:757).
 * Any output pointer may be NULL.  weights[n_rays, n_samples].
 */
typedef struct idn_composite_out {
    float* rgb_map;   /* [n,3] */
    float* disp_map;  /* [n]   */
    float* acc_map;   /* [n]   */
    float* depth_map; /* [n]   */
    float* weights;   /* [n,S] */
    float* rgb_fg;    /* [n,3] */
    float* last_weight; /* [n] = weights[:, -1] */
} idn_composite_out;

/* sigma_noise[n_rays, n_samples] (may be NULL) is added to the raw density before its ReLU: the caller draws
 * it as the reference does, randn * raw_noise_std (baseline.py:353-361).  white_bkgd != 0 adds 1 - acc_map to
 * rgb_map (:372-373).  The reference's Network never enables either (audio_exp_nerf.py:297-299). */
int idealnerf_composite_fwd(const float* raw, const float* z, const float* rays, const float* bc_rgb,
                            const float* sigma_noise, int white_bkgd, int64_t n_rays, int n_samples,
                            const idn_composite_out* out, void* stream);

/*
 * sample_pdf + merge (helper.py:269-313, audio_exp_nerf.py:340-349).
 *   weights[n, S] are the coarse compositing weights; the kernel uses weights[:, 1:-1]
 *   and bins = midpoints of z[n, S].
 *   u: [n_importance] shared by all rays when u_per_ray == 0 (deterministic,
 *      torch.linspace buffer), or [n, n_importance] when u_per_ray != 0.
 * Outputs (each may be NULL): z_samples[n, Ni], inds[n, Ni] (int64, searchsorted
 * right=True result), cdf[n, S-1], z_fine[n, S+Ni] (sorted union), z_std[n].
 * For identical weights in, cdf and inds are bit-identical to the reference's CPU path: the
 * normalising torch.sum is evaluated in ATen's own order (8 vector lanes, 4 accumulators), the
 * cumsum in fp64 rounded per element.
 */
int idealnerf_sample_pdf_fwd(const float* z, const float* weights, const float* u, int u_per_ray,
                             int64_t n_rays, int n_samples, int n_importance, float* z_samples, int64_t* inds,
                             float* cdf, float* z_fine, float* z_std, void* stream);

/*
 * The ray march between the two network passes as one kernel (audio_exp_nerf.py:335-349): raw2outputs of the
 * coarse pass, sample_pdf on its weights and the sorted merge with the coarse depths, one wavefront per ray.
 * Same arithmetic and outputs as idealnerf_composite_fwd followed by idealnerf_sample_pdf_fwd (bit for bit),
 * but the [n, S] weight matrix stays in LDS: out->weights is written only if it is not NULL.
 * idealnerf_render_rays_fwd uses it for every render with n_importance > 0.
 */
int idealnerf_march_fwd(const float* raw, const float* z, const float* rays, const float* bc_rgb, const float* sigma_noise,
                        int white_bkgd, const float* u, int u_per_ray, int64_t n_rays, int n_samples, int n_importance,
                        const idn_composite_out* out, float* z_samples, int64_t* inds, float* cdf, float* z_fine,
                        float* z_std, void* stream);

/* helper.sample_pdf with its own argument list (NeRFs/HeadNeRF/helper.py:269-313): bins[n, n_bins],
 * weights[n, n_bins-1] (the interior weights as the caller sliced them, before the +1e-5), u as above
 * -> z_samples[n, Ni], inds[n, Ni], cdf[n, n_bins] (each may be NULL).  Same kernel, same pdf / cdf
 * arithmetic as idealnerf_sample_pdf_fwd: torch.sum in ATen's CPU order, cumsum in fp64. */
int idealnerf_sample_pdf_bins_fwd(const float* bins, const float* weights, const float* u, int u_per_ray,
                                  int64_t n_rays, int n_bins, int n_importance, float* z_samples, int64_t* inds,
                                  float* cdf, void* stream);

/* The bit-exact boundary on its own (helper.py:297-310): given cdf[n, nb], bins[n, nb],
 * u (as above) -> inds (int64) and z_samples. */
int idealnerf_invert_cdf(const float* cdf, const float* bins, const float* u, int u_per_ray, int64_t n_rays,
                         int n_bins, int n_importance, float* z_samples, int64_t* inds, void* stream);

/*
 * Network.render_rays forward (audio_exp_nerf.py:297-371), all stages on `stream`.
 */
typedef struct idn_render_args {
    const float* rays;    /* [n,11] */
    const float* bc_rgb;  /* [n,3]  */
    int64_t n_rays;
    int n_samples;        /* N_samples   (2..256)  */
    int n_importance;     /* N_importance (0..256) */
    int precision;
    const float* packed_coarse;
    const float* folded_coarse;
    const float* packed_fine;   /* may be NULL when n_importance == 0 */
    const float* folded_fine;
    const float* t_vals;  /* [n_samples] torch.linspace(0,1,N_samples) */
    const float* t_rand;  /* [n, n_samples] or NULL (perturb == 0, or rng_mode == 1) */
    const float* u;       /* [n_importance] (u_per_ray=0) or [n, n_importance]; NULL with rng_mode == 1 */
    int u_per_ray;
    /* outputs, any may be NULL */
    float* rgb_map;  float* disp_map;  float* acc_map;  float* depth_map;  float* last_weight;  float* rgb_fg;
    float* rgb0;     float* disp0;     float* acc0;     float* z_std;      float* last_weight0; float* rgb_fg0;
    /* debug taps, any may be NULL */
    float* tap_z_coarse;   /* [n,S]     */
    float* tap_raw_coarse; /* [n,S,4]   */
    float* tap_weights_coarse; /* [n,S] */
    float* tap_cdf;        /* [n,S-1]   */
    int64_t* tap_inds;     /* [n,Ni]    */
    float* tap_z_samples;  /* [n,Ni]    */
    float* tap_z_fine;     /* [n,S+Ni]  */
    float* tap_raw_fine;   /* [n,S+Ni,4]*/
    float* tap_weights_fine; /* [n,S+Ni]*/
    void* workspace;       /* idealnerf_render_workspace_bytes() bytes */
    size_t workspace_bytes;
    /* Arithmetic of the FINE network's launches: 0 = same as `precision`, else IDN_PREC_* + 1.
     * "mixed" rendering = precision IDN_PREC_F32 with precision_fine_plus1 = IDN_PREC_BF16X3 + 1: the
     * coarse pass, whose output drives the importance sampling (which amplifies arithmetic noise), stays
     * exact; the fine pass (3/4 of the samples, nothing sampled after it) runs at the bf16 matrix rate.
     * packed_fine must have been packed for that arithmetic. */
    int precision_fine_plus1;
    /* render_rays' remaining switches (audio_exp_nerf.py:297-299; the reference leaves them at their defaults) */
    int lindisp;               /* coarse depths linear in inverse depth */
    int white_bkgd;            /* rgb += 1 - acc, in both passes */
    const float* noise_coarse; /* [n, n_samples] or NULL: raw_noise_std noise of the coarse pass, drawn by the caller */
    const float* noise_fine;   /* [n, n_samples + n_importance] or NULL */
    /* 0: the kernel sequence (network, march, network, compositing; raw outputs and fine depths through HBM).
     * 1: the whole per-ray path as ONE kernel that keeps a ray's sample positions, raw network outputs, weights and cdf in
     *    LDS -- nothing per-sample crosses HBM (csrc/render_fused.hip); 0-0.5 % faster than the sequence on a full frame, but it
     *    re-fetches a network's weight stream into every L2 at each change of network (six times the sequence's HBM bytes).
     * 2: the same kernel as two launches -- coarse network + march | fine network + compositing: one network per launch,
     *    so the weight stream stays in L2, and only the 768 bytes of fine depths per ray cross HBM between the launches
     *    (half the sequence's HBM bytes, 0.5 % slower).
     * 1 and 2 give the sequence's results bit for bit; they are built for the fp32 arithmetic at n_samples = 64,
     * n_importance = 128 without density noise (anything else returns IDN_EUNSUPPORTED).  DESIGN.md section 3. */
    int fused_march;
    /* The reference's perturb > 0 draws made inside the kernels instead of handed over as tensors ("optional
     * t_rand / u buffers or Philox seed").  rng_mode = 1: t_rand and u must be NULL; ray r of this call (counted from rng_ray0, so
     * that the chunks, row bands or ranks of one frame draw from ONE table whatever the partition) uses
     *   t_rand[r, s] = idealnerf_philox_uniform's value (which = 0, row rng_ray0 + r, column s)   -- torch.rand(z_vals.shape),
     *                                                                                                 audio_exp_nerf.py:314-326
     *   u[r, j]      = ... (which = 1, row rng_ray0 + r, column j)                                 -- torch.rand(.., N_samples),
     *                                                                                                 helper.py:283
     * in the kernels that consume them (coarse depths; inverse CDF of the march / the fused ray kernel): no [n, S] / [n, Ni]
     * random tensor exists.  Same distribution as torch.rand (24-bit uniform on [0, 1)); the NUMBERS are this library's own
     * (upstream's CUDA generator is no contract either: its numbers depend on its launch geometry). */
    int rng_mode;
    uint64_t rng_seed;
    int64_t rng_ray0;
} idn_render_args;

/* The table rng_mode = 1 draws from, as a tensor -- stands where upstream calls torch.rand for `t_rand` (audio_exp_nerf.py:314-326)
 * and `u` (helper.py:283); for the stand-alone entries, which take t_rand / u as tensors, and for checking:
 * out[r, c] (r < n_rows, c < n_cols) = the 24-bit uniform ((x >> 8) * 2^-24) of word c % 4 of Philox4x32-10 (Salmon et al., SC'11;
 * ten rounds, multipliers 0xD2511F53 / 0xCD9E8D57, key increments 0x9E3779B9 / 0xBB67AE85) with key = (seed low, seed high) and
 * counter = (c / 4, 0, low, high of 2 * (row0 + r) + which).  which: 0 = stratified offsets (t_rand), 1 = importance draws (u). */
int idealnerf_philox_uniform(uint64_t seed, int which, int64_t row0, int64_t n_rows, int n_cols, float* out, void* stream);

size_t idealnerf_render_workspace_bytes(int64_t n_rays, int n_samples, int n_importance);
int idealnerf_render_rays_fwd(const idn_render_args* a, void* stream);

/*
 * Full-frame mode: Network.render_dynamic_face with `render_poses` (audio_exp_nerf.py:396-427) -- get_rays
 * (NeRFs/HeadNeRF/helper.py:228-243; the principal point defaults to W/2, H/2 as at :402), the viewing directions and the
 * [o, d, near, far, viewdir] records (:407-427) followed by batchify_rays / render_rays -- as ONE call: the rays of the pixels
 * [row0 * W, (row0 + nrows) * W) are derived on the device from the camera, one internal pass (32 768 rays) at a time into a
 * 1.4 MB scratch that stays L2-resident; no [H * W, 11] ray tensor exists and nothing is scattered to a rank but its
 * (row0, nrows).  `a->rays` must be NULL, `a->n_rays` = nrows * W, `a->bc_rgb` the band's background pixels; everything else as
 * idealnerf_render_rays_fwd (same kernels: the records are the ones idealnerf_frame_rays writes, bit for bit).
 * `rays_out` (may be NULL): [n_rays, 11] copy of the records (debug tap).
 */
typedef struct idn_frame {
    float c2w[12];        /* row-major [3][4] camera-to-world */
    int H, W;
    float focal, cx, cy;  /* cx, cy < 0: W/2, H/2 */
    float near_, far_;
    int row0, nrows;
    float* rays_out;
} idn_frame;
size_t idealnerf_render_frame_workspace_bytes(int64_t n_rays, int n_samples, int n_importance);
int idealnerf_render_frame_fwd(const idn_render_args* a, const idn_frame* frame, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (NeRFs/HeadNeRF/train/audio_exp_nerf.py:534-552).  The forward of a pass is
 * idealnerf_coarse_depths / _query_rays_train_fwd / _composite_fwd / _sample_pdf_fwd; the
 * backward of a pass is one call.  z_samples are detached upstream (:345), so depths carry
 * no gradient; gradients reach the network parameters and the per-frame aud / latent vectors.
 * ------------------------------------------------------------------------------------------ */

/* Floats of the activation slab idealnerf_query_rays_train_fwd fills for n_points points
 * (2560 columns of saved activations + 88 floats of packed ReLU masks, x n_points rounded up to 128 rows;
 * need not be initialised). */
size_t idealnerf_train_acts_floats(int64_t n_points);

/* idealnerf_query_rays_fwd that also records what the backward needs (the post-ReLU
 * activations of all 11 hidden layers and both encodings).  precision: IDN_PREC_F32 or IDN_PREC_BF16X6 (the
 * fp32-grade arithmetics; `packed` must be that precision's stream).  idealnerf_pass_bwd itself computes the delta
 * chain and the 256 x 256 weight-gradient products with the six-piece bf16 arithmetic, or on
 * the fp32 matrix pipe when the process was started with IDN_BACKWARD_PIPE=f32. */
int idealnerf_query_rays_train_fwd(const float* packed, const float* folded, int precision, const float* rays,
                                   const float* z, int64_t n_rays, int n_samples, float* raw, float* acts,
                                   void* stream);

/* Gradient tensors, same shapes/layout as idn_facenerf_params (every entry is overwritten,
 * including the conditioning columns). */
typedef struct idn_facenerf_grads {
    float* pts_w[8];
    float* pts_b[8];
    float* views_w[3];
    float* views_b[3];
    float* alpha_w;
    float* alpha_b;
    float* rgb_w;
    float* rgb_b;
} idn_facenerf_grads;

size_t idealnerf_pass_bwd_workspace_bytes(int64_t n_rays, int n_samples);

/*
 * Backward of raw2outputs + run_network + FaceNeRF for one pass of n_rays x n_samples points.
 *   g_rgb_map [n,3], g_rgb_fg [n,3], g_last_weight [n], g_acc [n]: upstream gradients of the
 *   compositing outputs (any may be NULL = zero).
 *   d_aud [dim_aud], d_latent [dim_latent]: ACCUMULATED (+=); may be NULL.
 */
int idealnerf_pass_bwd(const idn_facenerf_params* p, const idn_facenerf_grads* grads, const float* aud,
                       const float* expr, const float* latent, const float* acts, const float* raw, const float* z,
                       const float* rays, const float* bc_rgb, int64_t n_rays, int n_samples, const float* g_rgb_map,
                       const float* g_rgb_fg, const float* g_last_weight, const float* g_acc, float* d_aud,
                       float* d_latent, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Test aid (no reference counterpart): one 256 x 256 weight-gradient product of the training step in isolation,
 *   dW[256][256] = delta[rows, :256]^T . acts[rows, :256],   db[256] = column sums of delta (may be NULL),
 * rows a positive multiple of 128, row pitches >= 256 floats.  pipe = IDN_DW_PIPE_BF16X6: as idealnerf_pass_bwd computes
 * the 256-wide layers (each fp32 operand as the exact sum of three bf16 pieces, six piece products per product on the bf16
 * matrix pipe, fp32 accumulate); IDN_DW_PIPE_F32: the same product on the fp32 matrix pipe.  tests/ compares both with fp64.
 */
#define IDN_DW_PIPE_BF16X6 0
#define IDN_DW_PIPE_F32 1
#define IDN_DW_PIPE_BF16X6_PASS 2   /* BF16X6 with the split count of a training pass (2 #CUs / 9 workgroups: each keeps its fp32
                                       accumulators over 1/56 of the rows on a 256-CU part), not one split per CU */
size_t idealnerf_dw_gemm_workspace_bytes(void);
int idealnerf_dw_gemm(const float* delta, int ld_delta, const float* acts, int ld_acts, int64_t rows, float* dW, float* db,
                      int pipe, void* workspace, size_t workspace_bytes, void* stream);

/*
 * AudioNet.forward (models/audio_net.py:43-69: the per-frame audio latent of audio_exp_nerf.py:258-259, or of the eight windows
 * under the attention smoother, :235-257) as ONE kernel, and its backward as one: windows [n, 16, 29] (win_size 16) ->
 * out [n, dim_aud].  Parameters in nn.Conv1d / nn.Linear layout ([C_out, C_in, 3] / [out, in], fp32, contiguous):
 * encoder_conv.{0,2,4,6} and encoder_fc1.{0,2}.  `saved` (NULL for inference) receives the 640 post-activations per window the
 * backward needs (idealnerf_audio_net_saved_floats).  The backward (n <= 8 windows) OVERWRITES every gradient buffer with the
 * sum over the windows; the windows themselves are data (no gradient).
 */
typedef struct idn_audio_net_params {
    const float* conv_w[4];
    const float* conv_b[4];
    const float* fc_w[2];
    const float* fc_b[2];
    int dim_aud;
} idn_audio_net_params;
typedef struct idn_audio_net_grads {
    float* conv_w[4];
    float* conv_b[4];
    float* fc_w[2];
    float* fc_b[2];
} idn_audio_net_grads;
size_t idealnerf_audio_net_saved_floats(int n_windows);
int idealnerf_audio_net_fwd(const idn_audio_net_params* p, const float* windows, int n_windows, float* out, float* saved,
                            void* stream);
int idealnerf_audio_net_bwd(const idn_audio_net_params* p, const idn_audio_net_grads* grads, const float* windows,
                            const float* saved, const float* d_out, int n_windows, void* stream);

/*
 * Frame tail.  to8b = `(255 * np.clip(x, 0, 1)).astype(np.uint8)` (NeRFs/HeadNeRF/helper.py:154) on
 * the device: rgb [n_pixels,3] fp32 -> out [n_pixels,3] u8, bit-identical to numpy for finite input;
 * swap_rb != 0 writes the channels in reverse order (the cv2.cvtColor the reference leaves
 * commented out, test/eval_aud_exp_nerf.py:490).  nonfinite_flag (device int, may be NULL) is
 * OR-ed with 1 if any input value is NaN/Inf: one flag per frame replaces the reference's
 * per-chunk isnan/isinf host syncs (train/audio_exp_nerf.py:367-369); such a value is written as 0.
 */
int idealnerf_to8b(const float* rgb, int64_t n_pixels, int swap_rb, uint8_t* out, int* nonfinite_flag, void* stream);

/*
 * Measurement aid (no reference counterpart): between begin and end, every launch of
 * the fused PE+MLP kernel is bracketed by HIP events on its own stream.  end()
 * synchronises those events and returns the summed kernel time, the number of launches
 * and the number of points (ray-samples) they evaluated.  Process-wide; not for
 * concurrent use.
 */
void idealnerf_profile_begin(void);
int idealnerf_profile_end(double* total_ms, int64_t* launches, int64_t* points);

/* The same measurement split by kernel family (arrays of IDN_PROF_KINDS entries each; any may be NULL):
 * the training step's roofline needs its three MFMA kernels separately.  For the dW GEMMs `points` counts
 * the points each launch contracted over (one launch per layer). */
#define IDN_PROF_MLP_FWD 0      /* fused PE + MLP forward (inference) */
#define IDN_PROF_MLP_FWD_SAVE 1 /* the same with saved activations (training forward) */
#define IDN_PROF_DELTA_CHAIN 2  /* fused backward delta chain */
#define IDN_PROF_DW_GEMM 3      /* dW = delta^T . activation GEMMs (+ bias column sums) on the fp32 matrix pipe */
#define IDN_PROF_DW_GEMM_X6 4   /* the 256 x 256 dW GEMMs: six bf16 piece products per fp32 product, fp32 accumulate */
#define IDN_PROF_MLP_FWD_SAVE_X6 5 /* training forward as six bf16 piece products per fp32 product (IDN_PREC_BF16X6) */
#define IDN_PROF_DELTA_CHAIN_X6 6  /* backward delta chain, the same arithmetic */
#define IDN_PROF_KINDS 7
int idealnerf_profile_end_kinds(double* total_ms, int64_t* launches, int64_t* points);

#ifdef __cplusplus
}
#endif
#endif /* IDEALNERF_H */
