// extern "C" surface of libidealnerf.so (declared in include/idealnerf.h).
#include "idn_internal.h"
#include <cstdarg>
#include <cstdio>

namespace idn {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_precision(int precision) {
    if (precision == IDN_PREC_F32 || precision == IDN_PREC_BF16X3 || precision == IDN_PREC_BF16 || precision == IDN_PREC_FP16X3 ||
        precision == IDN_PREC_BF16X6)
        return IDN_OK;
    return fail(IDN_EUNSUPPORTED, "precision %d is unknown (IDN_PREC_F32, IDN_PREC_BF16X3, IDN_PREC_BF16, IDN_PREC_FP16X3, IDN_PREC_BF16X6)", precision);
}
static int check_precision_train(int precision) {
    if (precision == IDN_PREC_F32 || precision == IDN_PREC_BF16X6) return IDN_OK;
    return fail(IDN_EUNSUPPORTED, "the training path runs fp32-grade arithmetic only: IDN_PREC_F32 or IDN_PREC_BF16X6 (got %d)", precision);
}

int launch_mlp(int precision, const float* packed, const float* folded, const float* x, const float* rays,
               const float* z, const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw,
               hipStream_t s) {
    if (n_points > 0x7fffffffLL) return fail(IDN_EUNSUPPORTED, "n_points %lld exceeds 2^31-1 per launch", (long long)n_points);
    if ((rays || pts) && n_samples < 1) return fail(IDN_EINVAL, "n_samples < 1");
    if (precision == IDN_PREC_BF16X3)
        return launch_mlp_bf16x3(packed, folded, x, rays, z, pts, dirs, n_points, n_samples, raw, s);
    if (precision == IDN_PREC_BF16)
        return launch_mlp_bf16(packed, folded, x, rays, z, pts, dirs, n_points, n_samples, raw, s);
    if (precision == IDN_PREC_FP16X3)
        return launch_mlp_fp16x3(packed, folded, x, rays, z, pts, dirs, n_points, n_samples, raw, s);
    if (precision == IDN_PREC_BF16X6)
        return launch_mlp_bf16x6(packed, folded, x, rays, z, pts, dirs, n_points, n_samples, raw, s);
    return launch_mlp_f32(packed, folded, x, rays, z, pts, dirs, n_points, n_samples, raw, s);
}

static int check_params(const idn_facenerf_params* p) {
    if (!p) return fail(IDN_EINVAL, "params is NULL");
    for (int i = 0; i < 8; ++i)
        if (!p->pts_w[i] || !p->pts_b[i]) return fail(IDN_EINVAL, "pts_linears.%d is NULL", i);
    for (int i = 0; i < 3; ++i)
        if (!p->views_w[i] || !p->views_b[i]) return fail(IDN_EINVAL, "views_linears.%d is NULL", i);
    if (!p->alpha_w || !p->alpha_b || !p->rgb_w || !p->rgb_b) return fail(IDN_EINVAL, "alpha/rgb head is NULL");
    if (p->dim_aud < 0 || p->dim_expr < 0 || p->dim_latent < 0) return fail(IDN_EINVAL, "negative conditioning width");
    return IDN_OK;
}

// ---- MLP launch timing (bench.py's roofline figure) ---------------------------------
// HIP events recorded on the launch stream itself, immediately around the kernel.
// Process-wide and meant for single-threaded measurement runs only.
static const int kProfSlots = 8192;
static bool g_prof_on = false;
static int g_prof_n = 0;
static hipEvent_t g_prof_ev[kProfSlots][2];
static int g_prof_created = 0;
static int64_t g_prof_points[kProfSlots];
static int g_prof_kind[kProfSlots];

ProfScope::ProfScope(hipStream_t s_, int64_t points, int kind) : slot(-1), s(s_) {
    if (!g_prof_on || g_prof_n >= kProfSlots) return;
    slot = g_prof_n++;
    if (slot >= g_prof_created) {
        (void)hipEventCreate(&g_prof_ev[slot][0]);
        (void)hipEventCreate(&g_prof_ev[slot][1]);
        g_prof_created = slot + 1;
    }
    g_prof_points[slot] = points;
    g_prof_kind[slot] = (kind >= 0 && kind < IDN_PROF_KINDS) ? kind : 0;
    (void)hipEventRecord(g_prof_ev[slot][0], s);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof_ev[slot][1], s);
}

}  // namespace idn

using namespace idn;

extern "C" {

int idealnerf_version(void) { return 4; }   // 4: idn_render_args grew by `rng_mode / rng_seed / rng_ray0` (round 4); 3: by `fused_march`
const char* idealnerf_last_error(void) { return g_err; }

size_t idealnerf_packed_weight_floats(int precision) {
    // 4 bytes per weight (fp32, or bf16 hi + bf16 lo); 2 for the plain-bf16 stream
    if (precision == IDN_PREC_F32 || precision == IDN_PREC_BF16X3 || precision == IDN_PREC_FP16X3) return (size_t)kStreamFrags * kFragFloats;
    if (precision == IDN_PREC_BF16) return (size_t)kPlainStreamFrags * kFragFloats;
    if (precision == IDN_PREC_BF16X6) return (size_t)kX6StreamFrags * kFragFloats;   // three bf16 pieces per weight, 48-fragment slices
    return 0;
}
size_t idealnerf_folded_bias_floats(void) { return kBiasFloats; }

int idealnerf_pack_weights(const idn_facenerf_params* p, int precision, float* packed, void* stream) {
    if (int e = check_params(p)) return e;
    if (int e = check_precision(precision)) return e;
    if (!packed) return fail(IDN_EINVAL, "packed is NULL");
    if (precision == IDN_PREC_BF16X3) return launch_pack_bf16x3(*p, packed, (hipStream_t)stream);
    if (precision == IDN_PREC_FP16X3) return launch_pack_bf16x3(*p, packed, (hipStream_t)stream, 1);
    if (precision == IDN_PREC_BF16) return launch_pack_bf16(*p, packed, (hipStream_t)stream);
    if (precision == IDN_PREC_BF16X6) return launch_pack_bf16x6(*p, packed, (hipStream_t)stream);
    return launch_pack_f32(*p, packed, (hipStream_t)stream);
}

int idealnerf_fold_conditioning(const idn_facenerf_params* p, const float* aud, const float* expr,
                                const float* latent, float* folded, void* stream) {
    if (int e = check_params(p)) return e;
    if (!folded) return fail(IDN_EINVAL, "folded is NULL");
    if ((p->dim_aud > 0) != (aud != nullptr)) return fail(IDN_EINVAL, "aud pointer does not match dim_aud=%d", p->dim_aud);
    if ((p->dim_expr > 0) != (expr != nullptr)) return fail(IDN_EINVAL, "expr pointer does not match dim_expr=%d", p->dim_expr);
    if ((p->dim_latent > 0) != (latent != nullptr))
        return fail(IDN_EINVAL, "latent pointer does not match dim_latent=%d", p->dim_latent);
    return launch_fold(*p, aud, expr, latent, folded, (hipStream_t)stream);
}

int idealnerf_facenerf_fwd(const float* packed, const float* folded, int precision, const float* x, int64_t n,
                           float* out, void* stream) {
    if (int e = check_precision(precision)) return e;
    if (n < 0) return fail(IDN_EINVAL, "n < 0");
    if (n == 0) return IDN_OK;
    if (!packed || !folded || !x || !out) return fail(IDN_EINVAL, "NULL pointer");
    return launch_mlp(precision, packed, folded, x, nullptr, nullptr, nullptr, nullptr, n, 1, out, (hipStream_t)stream);
}

int idealnerf_query_rays_fwd(const float* packed, const float* folded, int precision, const float* rays,
                             const float* z, int64_t n_rays, int n_samples, float* raw, void* stream) {
    if (int e = check_precision(precision)) return e;
    if (n_rays < 0 || n_samples < 1) return fail(IDN_EINVAL, "bad sizes n_rays=%lld n_samples=%d", (long long)n_rays, n_samples);
    if (n_rays == 0) return IDN_OK;
    if (!packed || !folded || !rays || !z || !raw) return fail(IDN_EINVAL, "NULL pointer");
    return launch_mlp(precision, packed, folded, nullptr, rays, z, nullptr, nullptr, n_rays * n_samples, n_samples, raw,
                      (hipStream_t)stream);
}

int idealnerf_query_points_fwd(const float* packed, const float* folded, int precision, const float* pts,
                               const float* viewdirs, int64_t n_rays, int n_samples, float* raw, void* stream) {
    if (int e = check_precision(precision)) return e;
    if (n_rays < 0 || n_samples < 1) return fail(IDN_EINVAL, "bad sizes n_rays=%lld n_samples=%d", (long long)n_rays, n_samples);
    if (n_rays == 0) return IDN_OK;
    if (!packed || !folded || !pts || !viewdirs || !raw) return fail(IDN_EINVAL, "NULL pointer");
    return launch_mlp(precision, packed, folded, nullptr, nullptr, nullptr, pts, viewdirs, n_rays * n_samples, n_samples,
                      raw, (hipStream_t)stream);
}

int idealnerf_frame_rays(const float* c2w, int H, int W, float focal, float cx, float cy, float near_, float far_,
                         int row0, int nrows, float* rays_out, void* stream) {
    if (!c2w || !rays_out) return fail(IDN_EINVAL, "NULL pointer");
    if (H <= 0 || W <= 0 || row0 < 0 || nrows < 0 || row0 + nrows > H) return fail(IDN_EINVAL, "bad frame/rows");
    if (nrows == 0) return IDN_OK;
    return launch_frame_rays(c2w, H, W, focal, cx, cy, near_, far_, row0, nrows, rays_out, (hipStream_t)stream);
}

int idealnerf_philox_uniform(uint64_t seed, int which, int64_t row0, int64_t n_rows, int n_cols, float* out, void* stream) {
    if (which != 0 && which != 1) return fail(IDN_EINVAL, "which %d (0 = stratified offsets, 1 = importance draws)", which);
    if (row0 < 0 || n_rows < 0 || n_cols < 0) return fail(IDN_EINVAL, "bad sizes");
    if (n_rows == 0 || n_cols == 0) return IDN_OK;
    if (!out) return fail(IDN_EINVAL, "NULL pointer");
    return launch_philox_uniform((unsigned long long)seed, which, row0, n_rows, n_cols, out, (hipStream_t)stream);
}

int idealnerf_coarse_depths(const float* rays, const float* t_vals, const float* t_rand, int lindisp, int64_t n_rays,
                            int n_samples, float* z, void* stream) {
    if (n_rays < 0 || n_samples < 1) return fail(IDN_EINVAL, "bad sizes");
    if (n_rays == 0) return IDN_OK;
    if (!rays || !t_vals || !z) return fail(IDN_EINVAL, "NULL pointer");
    return launch_coarse_depths(rays, t_vals, t_rand, n_rays, n_samples, lindisp, z, (hipStream_t)stream);
}

size_t idealnerf_audio_net_saved_floats(int n_windows) { return audio_net_saved_floats(n_windows); }
int idealnerf_audio_net_fwd(const idn_audio_net_params* p, const float* windows, int n_windows, float* out, float* saved,
                            void* stream) {
    return launch_audio_net_fwd(p, windows, n_windows, out, saved, (hipStream_t)stream);
}
int idealnerf_audio_net_bwd(const idn_audio_net_params* p, const idn_audio_net_grads* grads, const float* windows,
                            const float* saved, const float* d_out, int n_windows, void* stream) {
    return launch_audio_net_bwd(p, grads, windows, saved, d_out, n_windows, (hipStream_t)stream);
}

int idealnerf_to8b(const float* rgb, int64_t n_pixels, int swap_rb, uint8_t* out, int* nonfinite_flag, void* stream) {
    if (n_pixels < 0) return fail(IDN_EINVAL, "n_pixels < 0");
    if (n_pixels == 0) return IDN_OK;
    if (!rgb || !out) return fail(IDN_EINVAL, "NULL pointer");
    return launch_to8b(rgb, n_pixels, swap_rb, out, nonfinite_flag, (hipStream_t)stream);
}

int idealnerf_composite_fwd(const float* raw, const float* z, const float* rays, const float* bc_rgb,
                            const float* sigma_noise, int white_bkgd, int64_t n_rays, int n_samples,
                            const idn_composite_out* out, void* stream) {
    if (n_rays < 0) return fail(IDN_EINVAL, "n_rays < 0");
    if (n_rays == 0) return IDN_OK;
    if (!raw || !z || !rays || !bc_rgb || !out) return fail(IDN_EINVAL, "NULL pointer");
    return launch_composite(raw, z, rays, bc_rgb, n_rays, n_samples, sigma_noise, white_bkgd, *out, (hipStream_t)stream);
}

int idealnerf_sample_pdf_fwd(const float* z, const float* weights, const float* u, int u_per_ray, int64_t n_rays,
                             int n_samples, int n_importance, float* z_samples, int64_t* inds, float* cdf,
                             float* z_fine, float* z_std, void* stream) {
    if (n_rays < 0) return fail(IDN_EINVAL, "n_rays < 0");
    if (n_rays == 0) return IDN_OK;
    if (!z || !weights || !u) return fail(IDN_EINVAL, "NULL pointer");
    if (n_samples < 3) return fail(IDN_EUNSUPPORTED, "sample_pdf needs n_samples >= 3");
    return launch_sample_pdf(z, weights, nullptr, nullptr, u, u_per_ray, n_rays, n_samples, n_importance, z_samples,
                             inds, cdf, z_fine, z_std, (hipStream_t)stream);
}

int idealnerf_march_fwd(const float* raw, const float* z, const float* rays, const float* bc_rgb, const float* sigma_noise,
                        int white_bkgd, const float* u, int u_per_ray, int64_t n_rays, int n_samples, int n_importance,
                        const idn_composite_out* out, float* z_samples, int64_t* inds, float* cdf, float* z_fine,
                        float* z_std, void* stream) {
    if (n_rays < 0) return fail(IDN_EINVAL, "n_rays < 0");
    if (n_rays == 0) return IDN_OK;
    if (!raw || !z || !rays || !bc_rgb || !u || !out) return fail(IDN_EINVAL, "NULL pointer");
    return launch_march(raw, z, rays, bc_rgb, sigma_noise, white_bkgd, *out, u, u_per_ray, n_rays, n_samples, n_importance,
                        z_samples, inds, cdf, z_fine, z_std, (hipStream_t)stream);
}

int idealnerf_sample_pdf_bins_fwd(const float* bins, const float* weights, const float* u, int u_per_ray,
                                  int64_t n_rays, int n_bins, int n_importance, float* z_samples, int64_t* inds,
                                  float* cdf, void* stream) {
    if (n_rays < 0) return fail(IDN_EINVAL, "n_rays < 0");
    if (n_rays == 0) return IDN_OK;
    if (!bins || !weights || !u) return fail(IDN_EINVAL, "NULL pointer");
    return launch_sample_pdf(nullptr, weights, nullptr, bins, u, u_per_ray, n_rays, n_bins + 1, n_importance, z_samples,
                             inds, cdf, nullptr, nullptr, (hipStream_t)stream);
}

int idealnerf_invert_cdf(const float* cdf, const float* bins, const float* u, int u_per_ray, int64_t n_rays,
                         int n_bins, int n_importance, float* z_samples, int64_t* inds, void* stream) {
    if (n_rays < 0) return fail(IDN_EINVAL, "n_rays < 0");
    if (n_rays == 0) return IDN_OK;
    if (!cdf || !bins || !u) return fail(IDN_EINVAL, "NULL pointer");
    return launch_sample_pdf(nullptr, nullptr, cdf, bins, u, u_per_ray, n_rays, n_bins + 1, n_importance, z_samples,
                             inds, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

size_t idealnerf_train_acts_floats(int64_t n_points) {
    if (n_points <= 0) return 0;
    return (size_t)((n_points + 127) / 128 * 128) * kActColsAll;
}

int idealnerf_query_rays_train_fwd(const float* packed, const float* folded, int precision, const float* rays,
                                   const float* z, int64_t n_rays, int n_samples, float* raw, float* acts,
                                   void* stream) {
    if (int e = check_precision_train(precision)) return e;
    if (n_rays < 0 || n_samples < 1) return fail(IDN_EINVAL, "bad sizes");
    if (n_rays == 0) return IDN_OK;
    if (!packed || !folded || !rays || !z || !raw || !acts) return fail(IDN_EINVAL, "NULL pointer");
    const int64_t n = n_rays * n_samples;
    const int64_t p_pad = (n + 127) / 128 * 128;
    // (both activation-saving kernels write every row of the slab, the p_pad - n padding rows included -- they repeat the last
    //  point, and their deltas are zero: tests/test_hip_parity.py::test_train_forward_defines_every_row_of_the_activation_slab)
    if (precision == IDN_PREC_BF16X6)
        return launch_mlp_bf16x6(packed, folded, nullptr, rays, z, nullptr, nullptr, n, n_samples, raw, (hipStream_t)stream, acts, p_pad);
    return launch_mlp_f32(packed, folded, nullptr, rays, z, nullptr, nullptr, n, n_samples, raw, (hipStream_t)stream,
                          acts, p_pad);
}

size_t idealnerf_pass_bwd_workspace_bytes(int64_t n_rays, int n_samples) {
    if (n_rays <= 0 || n_samples <= 0) return 0;
    return bwd_workspace_bytes(n_rays * n_samples);
}

int idealnerf_pass_bwd(const idn_facenerf_params* p, const idn_facenerf_grads* grads, const float* aud,
                       const float* expr, const float* latent, const float* acts, const float* raw, const float* z,
                       const float* rays, const float* bc_rgb, int64_t n_rays, int n_samples, const float* g_rgb_map,
                       const float* g_rgb_fg, const float* g_last_weight, const float* g_acc, float* d_aud,
                       float* d_latent, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = check_params(p)) return e;
    if (!grads) return fail(IDN_EINVAL, "grads is NULL");
    for (int i = 0; i < 8; ++i)
        if (!grads->pts_w[i] || !grads->pts_b[i]) return fail(IDN_EINVAL, "grad pts_linears.%d is NULL", i);
    for (int i = 0; i < 3; ++i)
        if (!grads->views_w[i] || !grads->views_b[i]) return fail(IDN_EINVAL, "grad views_linears.%d is NULL", i);
    if (!grads->alpha_w || !grads->alpha_b || !grads->rgb_w || !grads->rgb_b) return fail(IDN_EINVAL, "grad head is NULL");
    if ((p->dim_aud > 0) != (aud != nullptr) || (p->dim_expr > 0) != (expr != nullptr) ||
        (p->dim_latent > 0) != (latent != nullptr))
        return fail(IDN_EINVAL, "conditioning pointers do not match the widths");
    if (n_rays < 0) return fail(IDN_EINVAL, "n_rays < 0");
    if (n_rays == 0) return IDN_OK;
    if (!acts || !raw || !z || !rays || !bc_rgb) return fail(IDN_EINVAL, "NULL pointer");
    return launch_pass_bwd(*p, *grads, aud, expr, latent, acts, raw, z, rays, bc_rgb, n_rays, n_samples, g_rgb_map,
                           g_rgb_fg, g_last_weight, g_acc, d_aud, d_latent, workspace, workspace_bytes,
                           (hipStream_t)stream);
}

size_t idealnerf_dw_gemm_workspace_bytes(void) { return dw_gemm_workspace_bytes(); }

int idealnerf_dw_gemm(const float* delta, int ld_delta, const float* acts, int ld_acts, int64_t rows, float* dW, float* db,
                      int pipe, void* workspace, size_t workspace_bytes, void* stream) {
    if (!delta || !acts || !dW) return fail(IDN_EINVAL, "NULL pointer");
    return launch_dw_gemm(delta, ld_delta, acts, ld_acts, rows, dW, db, pipe, workspace, workspace_bytes, (hipStream_t)stream);
}

void idealnerf_profile_begin(void) {
    g_prof_n = 0;
    g_prof_on = true;
}

int idealnerf_profile_end_kinds(double* total_ms, int64_t* launches, int64_t* points) {
    g_prof_on = false;
    double ms[IDN_PROF_KINDS] = {0};
    int64_t n[IDN_PROF_KINDS] = {0}, pts[IDN_PROF_KINDS] = {0};
    for (int i = 0; i < g_prof_n; ++i) {
        IDN_HIP_CHECK(hipEventSynchronize(g_prof_ev[i][1]));
        float t = 0;
        IDN_HIP_CHECK(hipEventElapsedTime(&t, g_prof_ev[i][0], g_prof_ev[i][1]));
        const int k = g_prof_kind[i];
        ms[k] += t;
        n[k] += 1;
        pts[k] += g_prof_points[i];
    }
    for (int k = 0; k < IDN_PROF_KINDS; ++k) {
        if (total_ms) total_ms[k] = ms[k];
        if (launches) launches[k] = n[k];
        if (points) points[k] = pts[k];
    }
    return IDN_OK;
}

int idealnerf_profile_end(double* total_ms, int64_t* launches, int64_t* points) {
    double ms[IDN_PROF_KINDS];
    int64_t n[IDN_PROF_KINDS], pts[IDN_PROF_KINDS];
    if (int e = idealnerf_profile_end_kinds(ms, n, pts)) return e;
    // the forward MLP launches, with or without saved activations
    if (total_ms) *total_ms = ms[IDN_PROF_MLP_FWD] + ms[IDN_PROF_MLP_FWD_SAVE] + ms[IDN_PROF_MLP_FWD_SAVE_X6];
    if (launches) *launches = n[IDN_PROF_MLP_FWD] + n[IDN_PROF_MLP_FWD_SAVE] + n[IDN_PROF_MLP_FWD_SAVE_X6];
    if (points) *points = pts[IDN_PROF_MLP_FWD] + pts[IDN_PROF_MLP_FWD_SAVE] + pts[IDN_PROF_MLP_FWD_SAVE_X6];
    return IDN_OK;
}

// ---- render_rays --------------------------------------------------------------------
static const int64_t kRenderChunk = 32768;  // rays per internal pass (bounds the workspace)

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RenderWs {
    float *z_c, *raw_c, *z_f, *raw_f;   // the compositing weights stay on chip (march kernel) or go straight to their tap
    size_t bytes;
};
static RenderWs carve(char* base, int64_t n, int S, int Ni) {
    RenderWs w;
    const int64_t c = n < kRenderChunk ? n : kRenderChunk;
    const int Sf = S + Ni;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = reinterpret_cast<float*>(base + off);
        off += align256(floats * sizeof(float));
        return p;
    };
    w.z_c = take((size_t)c * S);
    w.raw_c = take((size_t)c * S * 4);
    w.z_f = take((size_t)c * Sf);
    w.raw_f = take((size_t)c * Sf * 4);
    w.bytes = off;
    return w;
}

size_t idealnerf_render_workspace_bytes(int64_t n_rays, int n_samples, int n_importance) {
    if (n_rays <= 0 || n_samples <= 0 || n_importance < 0) return 0;
    return carve(nullptr, n_rays, n_samples, n_importance).bytes;
}

static int render_impl(const idn_render_args* a, const idn_frame* frame, void* stream_);
int idealnerf_render_rays_fwd(const idn_render_args* a, void* stream_) { return render_impl(a, nullptr, stream_); }

static size_t frame_scratch_bytes(int64_t n_rays) {
    const int64_t c = n_rays < kRenderChunk ? n_rays : kRenderChunk;
    return align256((size_t)c * IDN_RAY_FLOATS * sizeof(float));
}
size_t idealnerf_render_frame_workspace_bytes(int64_t n_rays, int n_samples, int n_importance) {
    const size_t base = idealnerf_render_workspace_bytes(n_rays, n_samples, n_importance);
    return base ? base + frame_scratch_bytes(n_rays) : 0;
}
int idealnerf_render_frame_fwd(const idn_render_args* a, const idn_frame* f, void* stream_) {
    if (!a || !f) return fail(IDN_EINVAL, "args / frame is NULL");
    if (a->rays) return fail(IDN_EINVAL, "frame mode derives the rays from the camera: args->rays must be NULL");
    if (f->H <= 0 || f->W <= 0 || f->row0 < 0 || f->nrows < 0 || f->row0 + f->nrows > f->H) return fail(IDN_EINVAL, "bad frame / rows");
    if (a->n_rays != (int64_t)f->nrows * f->W) return fail(IDN_EINVAL, "n_rays %lld != nrows * W = %lld", (long long)a->n_rays, (long long)f->nrows * f->W);
    return render_impl(a, f, stream_);
}

static int render_impl(const idn_render_args* a, const idn_frame* frame, void* stream_) {
    if (!a) return fail(IDN_EINVAL, "args is NULL");
    if (int e = check_precision(a->precision)) return e;
    const int prec_fine = a->precision_fine_plus1 ? a->precision_fine_plus1 - 1 : a->precision;
    if (int e = check_precision(prec_fine)) return e;
    const int64_t n = a->n_rays;
    const int S = a->n_samples, Ni = a->n_importance, Sf = S + Ni;
    if (n < 0 || S < 2 || Ni < 0) return fail(IDN_EINVAL, "bad sizes n=%lld S=%d Ni=%d", (long long)n, S, Ni);
    if (n == 0) return IDN_OK;
    if ((!a->rays && !frame) || !a->bc_rgb || !a->t_vals || !a->packed_coarse || !a->folded_coarse)
        return fail(IDN_EINVAL, "NULL input pointer");
    if (a->rng_mode != 0 && a->rng_mode != 1) return fail(IDN_EINVAL, "rng_mode %d (0 = t_rand / u as tensors, 1 = drawn in the kernels)", a->rng_mode);
    if (a->rng_mode && (a->t_rand || a->u)) return fail(IDN_EINVAL, "rng_mode 1 draws t_rand and u in the kernels: both pointers must be NULL");
    if (a->rng_mode && a->rng_ray0 < 0) return fail(IDN_EINVAL, "rng_ray0 < 0");
    if (Ni > 0 && (!a->packed_fine || !a->folded_fine || (!a->u && !a->rng_mode))) return fail(IDN_EINVAL, "fine pass inputs are NULL");
    if (Ni > 0 && S < 3) return fail(IDN_EUNSUPPORTED, "importance sampling needs n_samples >= 3");
    const size_t need_base = idealnerf_render_workspace_bytes(n, S, Ni);
    const size_t need = need_base + (frame ? frame_scratch_bytes(n) : 0);
    if (!a->workspace || a->workspace_bytes < need)
        return fail(IDN_EWORKSPACE, "workspace %zu bytes < required %zu", a->workspace_bytes, need);
    float* const ray_scratch = frame ? reinterpret_cast<float*>(reinterpret_cast<char*>(a->workspace) + need_base) : nullptr;
    if (a->fused_march != 0 && a->fused_march != 1 && a->fused_march != 2) return fail(IDN_EINVAL, "fused_march %d (0 = kernel sequence, 1 = one kernel, 2 = two kernels)", a->fused_march);
    if (a->fused_march) {
        if (a->precision != IDN_PREC_F32 || prec_fine != IDN_PREC_F32) return fail(IDN_EUNSUPPORTED, "fused march: built for the fp32 arithmetic");
        if (S != 64 || Ni != 128) return fail(IDN_EUNSUPPORTED, "fused march: built for n_samples = 64, n_importance = 128 (got %d, %d)", S, Ni);
        if (a->noise_coarse || a->noise_fine) return fail(IDN_EUNSUPPORTED, "fused march: no density noise");
    }
    hipStream_t st = (hipStream_t)stream_;
    const RenderWs w = carve(reinterpret_cast<char*>(a->workspace), n, S, Ni);

    auto off = [](auto* p, int64_t elems) { return p ? p + elems : p; };
    auto tap = [&](void* dst, const void* src, size_t bytes) -> int {
        if (!dst) return IDN_OK;
        IDN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return IDN_OK;
    };

    for (int64_t r0 = 0; r0 < n; r0 += kRenderChunk) {
        const int64_t c = (n - r0 < kRenderChunk) ? n - r0 : kRenderChunk;
        const float* rays = frame ? ray_scratch : a->rays + r0 * IDN_RAY_FLOATS;
        if (frame) {   // this pass's records: pixels [row0 W + r0, .. + c) of the frame, into the scratch the pass before has finished with
            if (int e = launch_frame_rays_pixels(frame->c2w, frame->H, frame->W, frame->focal, frame->cx, frame->cy, frame->near_, frame->far_,
                                                 (int64_t)frame->row0 * frame->W + r0, (int)c, ray_scratch, st))
                return e;
            if (frame->rays_out)
                IDN_HIP_CHECK(hipMemcpyAsync(frame->rays_out + r0 * IDN_RAY_FLOATS, ray_scratch, (size_t)c * IDN_RAY_FLOATS * sizeof(float),
                                             hipMemcpyDeviceToDevice, st));
        }
        const float* bc = a->bc_rgb + r0 * 3;
        const Draws draws{(unsigned long long)a->rng_seed, (long)(a->rng_ray0 + r0), a->rng_mode};   // this pass's rows of the draw table
        if (int e = launch_coarse_depths(rays, a->t_vals, off(a->t_rand, r0 * S), c, S, a->lindisp, w.z_c, st, draws)) return e;
        if (!a->fused_march)
            if (int e = launch_mlp(a->precision, a->packed_coarse, a->folded_coarse, nullptr, rays, w.z_c, nullptr, nullptr, c * S, S, w.raw_c, st)) return e;
        idn_composite_out co = {};
        const bool fine = Ni > 0;
        co.rgb_map = off(fine ? a->rgb0 : a->rgb_map, r0 * 3);
        co.disp_map = off(fine ? a->disp0 : a->disp_map, r0);
        co.acc_map = off(fine ? a->acc0 : a->acc_map, r0);
        co.depth_map = fine ? nullptr : off(a->depth_map, r0);
        co.weights = off(a->tap_weights_coarse, r0 * S);
        co.rgb_fg = off(fine ? a->rgb_fg0 : a->rgb_fg, r0 * 3);
        co.last_weight = off(fine ? a->last_weight0 : a->last_weight, r0);
        if (int e = tap(off(a->tap_z_coarse, r0 * S), w.z_c, (size_t)c * S * 4)) return e;
        if (a->fused_march) {   // one kernel from the coarse depths to the pixels; raw, weights, cdf and the fine depths stay in LDS
            idn_composite_out fo = {};
            fo.rgb_map = off(a->rgb_map, r0 * 3);
            fo.disp_map = off(a->disp_map, r0);
            fo.acc_map = off(a->acc_map, r0);
            fo.depth_map = off(a->depth_map, r0);
            fo.weights = off(a->tap_weights_fine, r0 * Sf);
            fo.rgb_fg = off(a->rgb_fg, r0 * 3);
            fo.last_weight = off(a->last_weight, r0);
            const float* u = a->u_per_ray ? off(a->u, r0 * Ni) : a->u;
            const bool split = a->fused_march == 2;   // two launches: the fine depths cross HBM (w.z_f), everything else stays on chip
            if (int e = launch_render_fused(a->fused_march, a->packed_coarse, a->folded_coarse, a->packed_fine, a->folded_fine, rays, bc, w.z_c,
                                            split ? w.z_f : nullptr, u, a->u_per_ray, c, a->white_bkgd, co, fo, off(a->z_std, r0),
                                            off(a->tap_raw_coarse, r0 * S * 4), off(a->tap_raw_fine, r0 * Sf * 4),
                                            split ? nullptr : off(a->tap_z_fine, r0 * Sf), off(a->tap_inds, r0 * Ni), off(a->tap_z_samples, r0 * Ni),
                                            off(a->tap_cdf, r0 * (S - 1)), st, draws))
                return e;
            if (split)
                if (int e = tap(off(a->tap_z_fine, r0 * Sf), w.z_f, (size_t)c * Sf * 4)) return e;
            continue;
        }
        if (int e = tap(off(a->tap_raw_coarse, r0 * S * 4), w.raw_c, (size_t)c * S * 16)) return e;
        if (!fine) {
            if (int e = launch_composite(w.raw_c, w.z_c, rays, bc, c, S, off(a->noise_coarse, r0 * S), a->white_bkgd, co, st)) return e;
            continue;
        }
        // the march between the passes: coarse compositing, inverse-CDF sampling and the merge in one kernel
        const float* u = a->u_per_ray ? off(a->u, r0 * Ni) : a->u;
        if (int e = launch_march(w.raw_c, w.z_c, rays, bc, off(a->noise_coarse, r0 * S), a->white_bkgd, co, u, a->u_per_ray, c, S, Ni,
                                 off(a->tap_z_samples, r0 * Ni), off(a->tap_inds, r0 * Ni), off(a->tap_cdf, r0 * (S - 1)),
                                 w.z_f, off(a->z_std, r0), st, draws))
            return e;
        if (int e = launch_mlp(prec_fine, a->packed_fine, a->folded_fine, nullptr, rays, w.z_f, nullptr, nullptr, c * Sf, Sf, w.raw_f, st)) return e;
        idn_composite_out fo = {};
        fo.rgb_map = off(a->rgb_map, r0 * 3);
        fo.disp_map = off(a->disp_map, r0);
        fo.acc_map = off(a->acc_map, r0);
        fo.depth_map = off(a->depth_map, r0);
        fo.weights = off(a->tap_weights_fine, r0 * Sf);
        fo.rgb_fg = off(a->rgb_fg, r0 * 3);
        fo.last_weight = off(a->last_weight, r0);
        if (int e = launch_composite(w.raw_f, w.z_f, rays, bc, c, Sf, off(a->noise_fine, r0 * Sf), a->white_bkgd, fo, st)) return e;
        if (int e = tap(off(a->tap_z_fine, r0 * Sf), w.z_f, (size_t)c * Sf * 4)) return e;
        if (int e = tap(off(a->tap_raw_fine, r0 * Sf * 4), w.raw_f, (size_t)c * Sf * 16)) return e;
    }
    return IDN_OK;
}

}  // extern "C"
