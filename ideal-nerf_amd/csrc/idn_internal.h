// Internal declarations shared by the HIP translation units of libidealnerf.so.
// gfx950 only.  See DESIGN.md for the data layout this header encodes.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <stddef.h>
#include "../../include/idealnerf.h"

namespace idn {

// ---------------------------------------------------------------------------
// Packed weight stream (fp32 path).
//
// The per-point part of one FaceNeRF is a flat sequence of 1 KiB "fragments" in the
// exact order the kernel consumes them.  Fragment (layer, g, t) holds, for lane
// l = (i = l & 31, h = l >> 5), the four floats
//      W_layer[n = 32 t + i][k = 8 g + 4 h + j],  j = 0..3
// i.e. the A operands of four consecutive v_mfma_f32_32x32x2_f32 whose B operands
// are registers 4q..4q+3 (q = g & 3) of accumulator tile T = g >> 2 of the previous
// layer: an accumulator register r of tile T holds, in lane half h, output channel
// 32 T + (r & 3) + 8 (r >> 2) + 4 h = 8 g + 4 h + j.  Activations therefore never
// leave registers between layers.
//
// Within a layer, fragments are ordered tile-major (all k-groups of n-tile 0, then tile 1,
// ...), so output tiles finish one after the other and their ReLU / the next tile's bias load
// ride in the MFMA shadow of the neighbouring tile.  K sources are concatenated in
// k-groups of 8 channels (zero padded): PE(63 -> 8 groups), hidden (256 -> 32 groups),
// direction PE (27 -> 4 groups).  The conditioning columns are not in the stream; they
// are folded into the per-frame bias block (idealnerf_fold_conditioning).
// ---------------------------------------------------------------------------
constexpr int kFragBytes = 1024;
constexpr int kFragFloats = 256;
// Ring geometry.  Measured on the bf16x3 kernel (512^2 frame): 2 x 64 KiB 4.12e8 samples/s,
// 4 x 32 KiB (prefetch three slices ahead, counted vmcnt) 3.98e8: the extra barriers cost more
// than the deeper prefetch buys; the stream's cost is its issue/bandwidth (timing-only build
// without it: 4.70e8), not its latency.
constexpr int kSliceFrags = 64;                 // one LDS ring slot = 64 KiB
constexpr int kRingSlots = 2;                   // slices are fetched kRingSlots-1 ahead of their use
constexpr int kRingFrags = kRingSlots * kSliceFrags;  // 128 KiB ring
constexpr int kSliceBytes = kSliceFrags * kFragBytes;

constexpr int kNumLayers = 12;  // pts0..7, views0(+alpha), views1, views2, rgb
// n-tiles (of 32 output channels) and k-groups (of 8 input channels) per layer
constexpr int kLayerNT[kNumLayers] = {8, 8, 8, 8, 8, 8, 8, 8, 5, 4, 4, 1};
constexpr int kLayerKG[kNumLayers] = {8, 32, 32, 32, 32, 40, 32, 32, 36, 16, 16, 16};

constexpr int layer_f0(int l) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += kLayerNT[i] * kLayerKG[i];
    return f;
}
constexpr int kUsedFrags = layer_f0(kNumLayers);                                  // 2244
constexpr int kNumSlices =
    ((kUsedFrags + kSliceFrags - 1) / kSliceFrags + kRingSlots - 1) / kRingSlots * kRingSlots;  // 36
constexpr int kStreamFrags = kNumSlices * kSliceFrags;                            // 2304
static_assert(kUsedFrags == 2244, "layer table changed");
static_assert(kNumSlices % kRingSlots == 0, "slot of a slice must be static across passes");
static_assert(kRingFrags == 128, "FragReader addresses the ring as two 64-fragment halves");

// Plain-bf16 stream (IDN_PREC_BF16): weights as one bf16 each, one fragment per 16-channel
// k-step, so every layer has half the fragments of the 4-byte streams above.
constexpr int plain_f0(int l) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += kLayerNT[i] * (kLayerKG[i] / 2);
    return f;
}
constexpr int kPlainUsedFrags = plain_f0(kNumLayers);                              // 1122
constexpr int kPlainNumSlices =
    ((kPlainUsedFrags + kSliceFrags - 1) / kSliceFrags + kRingSlots - 1) / kRingSlots * kRingSlots;  // 18
constexpr int kPlainStreamFrags = kPlainNumSlices * kSliceFrags;                   // 1152
static_assert(kPlainUsedFrags == 1122 && kPlainUsedFrags % 2 == 0, "plain stream table changed");

// Six-piece bf16 stream (IDN_PREC_BF16X6): per (n-tile, 16-channel k-step) a TRIPLE of fragments (p1, p2, p3) -- the three
// bf16 pieces of each weight.  Its ring slots hold 48 fragments = 16 k-steps, so slices, layer starts and ring phases sit on
// the same k-steps as in the 64-fragment rings above (16 k-steps of two fragments), and the stream is 3.375 MiB: it stays
// in a 4 MiB per-XCD L2.  (Round 2 padded every triple to a quad with a zero fragment: 4.5 MiB, 5 GB of fabric reads per
// launch where the other streams cause 43 MB.)
constexpr int kX6KFrags = 3;                                   // fragments per k-step
constexpr int kX6SliceFrags = 16 * kX6KFrags;                  // 48: one ring slot = 16 k-steps = 48 KiB
constexpr int kX6RingFrags = kRingSlots * kX6SliceFrags;       // 96 KiB ring
constexpr int kX6UsedFrags = kX6KFrags * kPlainUsedFrags;      // 3366
constexpr int kX6NumSlices = 2 * kNumSlices;                   // 72 (the same k-steps per slice as a quad stream had)
constexpr int kX6StreamFrags = kX6NumSlices * kX6SliceFrags;   // 3456 = 3.375 MiB
static_assert(kX6UsedFrags <= kX6StreamFrags && kX6StreamFrags - kX6UsedFrags < kX6RingFrags + kX6SliceFrags, "x6 stream table");

// Folded bias block: one float per output channel, natural channel order
// (accumulator register 4q+j of tile t, lane half h <-> channel 32 t + 8 q + 4 h + j).
constexpr int bias_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += kLayerNT[i] * 32;
    return o;
}
// ... followed by a copy of alpha_linear's weight row (256 floats): the fp32 kernel takes sigma as a
// 256-term dot product on the vector unit instead of a fifth 32-row MFMA tile of views_linears.0
// of which one row is used (1.6 % of the pass's MFMAs).
// and of rgb_linear's three weight rows (3 x 128): the same for the colour head (3 rows of 32 used).
constexpr int kAlphaOff = bias_off(kNumLayers);    // 2496
constexpr int kRgbOff = kAlphaOff + 256;           // 2752
constexpr int kBiasFloats = kRgbOff + 3 * 128;     // 3136
static_assert(kBiasFloats == 3136, "bias table changed");

// views0 carries sigma as channel 128 (tile 4, row 0): alpha_linear rides in the same
// pass over the trunk output instead of a separate N=1 layer.
constexpr int kSigmaChannel = 128;

// ---------------------------------------------------------------------------
// Training-mode activation slab: layer-major row-major matrices of p_pad rows each
// (p_pad = n_points rounded up to 128; rows beyond n_points stay zero).
//   x0  [p_pad, 64]   gamma10(point), column 63 zero
//   dir [p_pad, 64]   gamma4(view dir) in columns 0..26, rest zero
//   a1..a8 [p_pad, 256]  post-ReLU outputs of pts_linears.0..7
//   v1..v3 [p_pad, 128]  post-ReLU outputs of views_linears.0..2
// act_off(i) = first column of matrix i when the slab is viewed as 2560 columns.
// ---------------------------------------------------------------------------
enum { kActX0 = 0, kActDir = 1, kActA1 = 2, kActV1 = 10, kActCount = 13 };
constexpr int act_width(int i) { return i < 2 ? 64 : (i < kActV1 ? 256 : 128); }
constexpr int act_off(int i) {
    int o = 0;
    for (int k = 0; k < i; ++k) o += act_width(k);
    return o;
}
constexpr int kActCols = act_off(kActCount);  // 2560
static_assert(kActCols == 2560, "activation slab layout changed");
// Behind the matrices: the ReLU masks of the 11 hidden layers as BITS, in the accumulator layout both MLP
// kernels share, so the backward delta chain loads one uint4 per lane and layer instead of gathering 16-byte
// pieces of the saved activations.  Layer id 0..7 = a1..a8, 8..10 = v1..v3; entry
// [id][wave tile = point / 32][lane] is a uint4 whose dword k holds tiles 2k, 2k+1: bit 31 - (16 (T & 1) + r)
// = 1 where the PRE-activation of register r of tile T is <= 0 (the unit is off; +0.0 counts as off, as in torch's relu backward).
constexpr int kMaskLayers = 11;
constexpr int kMaskFloatsPerPoint = kMaskLayers * 8;   // 2 lanes x 4 dwords per point and layer
constexpr int kActColsAll = kActCols + kMaskFloatsPerPoint;
__host__ __device__ inline size_t mask_index(int id, int64_t p_pad, int64_t wave_tile, int lane) {
    return ((size_t)id * (size_t)(p_pad / 32) + (size_t)wave_tile) * 64 + (size_t)lane;   // in uint4 units
}

// ---------------------------------------------------------------------------
// error plumbing (capi.hip)
// ---------------------------------------------------------------------------
int fail(int code, const char* fmt, ...);
#define IDN_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return ::idn::fail(IDN_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---------------------------------------------------------------------------
// Per-device launch state of a kernel family: the CU count (grid = min(tiles, CUs)) and the
// dynamic-LDS opt-in, done once per device under a lock.  The reference may run its model under
// nn.DataParallel (one host thread per device in ONE process): launches are
// re-entrant per device and stream.
// ---------------------------------------------------------------------------
struct LaunchSetup {
    static constexpr int kMaxDevices = 64;
    std::mutex mu;
    int cus[kMaxDevices] = {};
    template <class OptIn>
    int get(OptIn&& opt_in, int* num_cu) {
        int dev = 0;
        IDN_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= kMaxDevices) return ::idn::fail(IDN_EUNSUPPORTED, "device index %d out of range", dev);
        std::lock_guard<std::mutex> lock(mu);
        if (!cus[dev]) {
            hipDeviceProp_t prop;
            IDN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
            if (int e = opt_in()) return e;
            cus[dev] = prop.multiProcessorCount;
        }
        *num_cu = cus[dev];
        return IDN_OK;
    }
};

// ---------------------------------------------------------------------------
// optional per-launch timing of the MLP kernel (capi.hip); off by default
// ---------------------------------------------------------------------------
struct ProfScope {
    int slot;
    hipStream_t s;
    ProfScope(hipStream_t s, int64_t points, int kind = IDN_PROF_MLP_FWD);
    ~ProfScope();
};

// ---------------------------------------------------------------------------
// launchers (one per .hip file)
// ---------------------------------------------------------------------------
int launch_pack_f32(const idn_facenerf_params& p, float* packed, hipStream_t s);
int launch_pack_bf16x3(const idn_facenerf_params& p, float* packed, hipStream_t s, int fmt = 0);  // fmt 1: fp16 halves
int launch_pack_bf16x6(const idn_facenerf_params& p, float* packed, hipStream_t s);
int launch_mlp_bf16x6(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                      const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s,
                      float* acts = nullptr, int64_t p_pad = 0);   // acts != null: the training forward (saves activations + ReLU masks)
int launch_pack_bf16(const idn_facenerf_params& p, float* packed, hipStream_t s);
int launch_mlp_bf16(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                    const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s);
int launch_mlp_bf16x3(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                      const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s);
int launch_mlp_fp16x3(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                      const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s);
// precision dispatch for the inference forward
int launch_mlp(int precision, const float* packed, const float* folded, const float* x, const float* rays,
               const float* z, const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw,
               hipStream_t s);
int launch_fold(const idn_facenerf_params& p, const float* aud, const float* expr, const float* latent,
                float* folded, hipStream_t s);
// x != nullptr: pre-embedded rows [n_points, 90]; pts != nullptr: raw points [n_points,3] +
// dirs[n_points/S, 3]; else rays[n_rays,11] + z[n_rays,S]
int launch_mlp_f32(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                   const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s,
                   float* acts = nullptr, int64_t p_pad = 0);

int launch_frame_rays(const float* c2w, int H, int W, float focal, float cx, float cy, float near_, float far_,
                      int row0, int nrows, float* rays_out, hipStream_t s);
int launch_frame_rays_pixels(const float* c2w_host, int H, int W, float focal, float cx, float cy, float near_, float far_,
                             int64_t pix0, int npix, float* rays_out, hipStream_t s);
size_t audio_net_saved_floats(int n);
int launch_audio_net_fwd(const idn_audio_net_params* p, const float* windows, int n, float* out, float* saved, hipStream_t s);
int launch_audio_net_bwd(const idn_audio_net_params* p, const idn_audio_net_grads* g, const float* windows, const float* saved,
                         const float* d_out, int n, hipStream_t s);
int launch_to8b(const float* rgb, int64_t n_pixels, int swap_rb, unsigned char* out, int* flag, hipStream_t s);
// In-kernel draws (include/idealnerf.h: rng_mode): row `ray0 + r` of the Philox table replaces t_rand[r, :] / u[r, :]
struct Draws {
    unsigned long long seed;
    long ray0;
    int on;
};
int launch_philox_uniform(unsigned long long seed, int which, int64_t row0, int64_t n_rows, int n_cols, float* out, hipStream_t s);
int launch_coarse_depths(const float* rays, const float* t_vals, const float* t_rand, int64_t n_rays, int S,
                         int lindisp, float* z, hipStream_t s, Draws draws = Draws{0, 0, 0});
int launch_composite(const float* raw, const float* z, const float* rays, const float* bc, int64_t n_rays, int S,
                     const float* noise, int white_bkgd, const idn_composite_out& out, hipStream_t s);
// coarse raw2outputs + sample_pdf + merge in one kernel (the weights stay on chip)
int launch_march(const float* raw, const float* z, const float* rays, const float* bc, const float* noise, int white_bkgd,
                 const idn_composite_out& out, const float* u, int u_per_ray, int64_t n_rays, int S, int Ni,
                 float* z_samples, int64_t* inds, float* cdf_out, float* z_fine, float* z_std, hipStream_t s,
                 Draws draws = Draws{0, 0, 0});
// render_fused.hip (fp32, S = 64, Ni = 128).  arrangement 1: coarse network -> march -> fine network -> compositing in ONE kernel;
// 2: coarse network + march | fine network + compositing (two launches, the fine depths z_f[n, 192] cross HBM between them)
int launch_render_fused(int arrangement, const float* packed_c, const float* folded_c, const float* packed_f, const float* folded_f,
                        const float* rays, const float* bc, const float* z_c, float* z_f, const float* u, int u_per_ray, int64_t n_rays,
                        int white_bkgd, const idn_composite_out& co, const idn_composite_out& fo, float* z_std, float* tap_raw_c,
                        float* tap_raw_f, float* tap_z_fine, int64_t* tap_inds, float* tap_z_samples, float* tap_cdf, hipStream_t s,
                        Draws draws = Draws{0, 0, 0});
int launch_sample_pdf(const float* z, const float* weights, const float* cdf_in, const float* bins_in,
                      const float* u, int u_per_ray, int64_t n_rays, int S, int Ni, float* z_samples,
                      int64_t* inds, float* cdf_out, float* z_fine, float* z_std, hipStream_t s);

// train.hip: backward of one render pass
// ---------------------------------------------------------------------------
// Backward delta chain (mlp_f32_bwd.hip): the transposed weights as a second fragment stream.
// Stage s computes D_in^T = W_s^T . D_out^T for 32 points per wave, masks it with the saved
// activation and keeps it in registers as the next stage's B operand.
//   0 rgb_linear^T (K 3, padded)   1 views_linears.2^T   2 views_linears.1^T
//   3 views_linears.0[:, :256]^T + alpha_linear^T as k-channel 128 (K 129, padded)
//   4..10 pts_linears.7 .. pts_linears.1 ^T (pts_linears.5: its 256 hidden columns)
// Stages 0-3 end at fragment 280; the stream is padded to 384 so that every 256-fragment trunk
// stage starts on the same ring phase (one loop body serves them all).
// ---------------------------------------------------------------------------
constexpr int kBwdStages = 11;
constexpr int kBwdNT[kBwdStages] = {4, 4, 4, 8, 8, 8, 8, 8, 8, 8, 8};
constexpr int kBwdKG[kBwdStages] = {2, 16, 16, 18, 32, 32, 32, 32, 32, 32, 32};
constexpr int kBwdHeadFrags = 280, kBwdTrunk0 = 384;
constexpr int bwd_f0(int s) {
    if (s >= 4) return kBwdTrunk0 + 256 * (s - 4);
    int f = 0;
    for (int i = 0; i < s; ++i) f += kBwdNT[i] * kBwdKG[i];
    return f;
}
static_assert(bwd_f0(3) + kBwdNT[3] * kBwdKG[3] == kBwdHeadFrags, "head stages");
constexpr int kBwdStreamFrags = kBwdTrunk0 + 7 * 256;   // 2176
static_assert(kBwdStreamFrags % kSliceFrags == 0 && kBwdTrunk0 % kRingFrags == 0, "ring phase of the trunk stages");
constexpr int kBwdNumSlices = kBwdStreamFrags / kSliceFrags;

int launch_pack_f32_bwd(const idn_facenerf_params& p, float* packed_bwd, hipStream_t s);
// d_rgb: [p_pad, 64] (cols 0..2 = d raw rgb), dv0: [p_pad, 256] (col 128 = d raw sigma; cols 0..127 are written),
// acts: the forward's activation slab, dv2/dv1: [p_pad, 128], da[l]: [p_pad, 256] = delta of pts_linears.l
int launch_delta_chain(const float* packed_bwd, const float* acts, int64_t p_pad, const float* d_rgb, float* dv0,
                       float* dv2, float* dv1, float* const da[8], hipStream_t s);
// the same chain as six bf16 piece products per fp32 product (its own transposed stream, twice the fragments)
size_t bwd_stream_floats_x6();
int launch_pack_bf16x6_bwd(const idn_facenerf_params& p, float* packed_bwd, hipStream_t s);
int launch_delta_chain_x6(const float* packed_bwd, const float* acts, int64_t p_pad, const float* d_rgb, float* dv0,
                          float* dv2, float* dv1, float* const da[8], hipStream_t s);
size_t bwd_workspace_bytes(int64_t n_points);
size_t dw_gemm_workspace_bytes();
int launch_dw_gemm(const float* delta, int ld_delta, const float* acts, int ld_acts, int64_t rows, float* dW, float* db, int pipe,
                   void* ws, size_t ws_bytes, hipStream_t s);
int launch_pass_bwd(const idn_facenerf_params& p, const idn_facenerf_grads& gr, const float* aud, const float* expr,
                    const float* latent, const float* acts, const float* raw, const float* z, const float* rays,
                    const float* bc, int64_t n_rays, int S, const float* g_rgb, const float* g_fg, const float* g_lw,
                    const float* g_acc, float* d_aud, float* d_latent, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace idn
