// Load-time and per-frame preparation kernels (gfx950):
//   pack:  state_dict tensors -> MFMA-fragment weight stream (layout: idn_internal.h)
//   fold:  per-frame conditioning vectors -> bias block
// Together they are the part of FaceNeRF.forward (models/face_nerf.py:41-55,58,61,68-70)
// that is constant per frame: the broadcast [aud | expr/3 | latent] columns of layers 0 and 5
// and the expr/3 columns of views_linears.0 contribute W[:, cols] . vector, a bias.
#include "idn_internal.h"
#include <hip/hip_fp16.h>

namespace idn {

struct PackLayer {
    const float* w;        // [rows, ld] nn.Linear weight
    const float* w_extra;  // optional single row appended as channel `rows_extra_at` (alpha_linear)
    int ld;
    int rows;              // valid output channels taken from w
    int extra_at;          // channel index of the extra row
    int f0, nt, kg, kg0;   // stream position; k-groups [0,kg0) read source 0, the rest source 1
    int col0[2];           // first column of each source inside w
    int kvalid[2];         // channels present in each source (rest of the k-group is zero)
};
struct PackDesc {
    PackLayer L[kNumLayers];
};

__global__ void pack_f32_kernel(PackDesc d, float4* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= kStreamFrags * 64) return;
    const int f = gid >> 6, lane = gid & 63;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (f < kUsedFrags) {
        int l = 0;
        while (l + 1 < kNumLayers && f >= d.L[l + 1].f0) ++l;
        const PackLayer& L = d.L[l];
        const int rel = f - L.f0;
        const int t = rel / L.kg, g = rel - t * L.kg;  // tile-major: a tile's k-groups are consecutive
        const int n = 32 * t + (lane & 31);
        const int src = g < L.kg0 ? 0 : 1;
        const int kbase = 8 * (src ? g - L.kg0 : g) + 4 * (lane >> 5);
        for (int j = 0; j < 4; ++j) {
            const int k = kbase + j;
            if (k < L.kvalid[src]) {
                if (n < L.rows) v[j] = L.w[(long)n * L.ld + L.col0[src] + k];
                else if (L.w_extra && n == L.extra_at && src == 0) v[j] = L.w_extra[k];
            }
        }
    }
    out[gid] = make_float4(v[0], v[1], v[2], v[3]);
}

// bf16x3 stream: fragment pair (2p, 2p+1) of a tile = hi / lo halves of one 16-channel k-step.
// Lane (i, h) holds 8 bf16: element j = W[32t+i][channel 16 ks + (j&3) + 8 (j>>2) + 4 h]
// (the order in which an accumulator tile's registers 8s..8s+7 become a B operand).
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;  // finite inputs only (weights)
}
// fmt 0: bf16 halves (round to nearest even); fmt 1: fp16 halves (IDN_PREC_FP16X3)
__device__ __forceinline__ unsigned f16_rne(float x) { return (unsigned)__half_as_ushort(__float2half_rn(x)); }
__device__ __forceinline__ float f16_to_f32(unsigned b) { return __half2float(__ushort_as_half((unsigned short)b)); }
__global__ void pack_bf16x3_kernel(PackDesc d, uint4* out, int fmt) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= kStreamFrags * 64) return;
    const int f = gid >> 6, lane = gid & 63;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (f < kUsedFrags) {
        int l = 0;
        while (l + 1 < kNumLayers && f >= d.L[l + 1].f0) ++l;
        const PackLayer& L = d.L[l];
        const int rel = f - L.f0;
        const int t = rel / L.kg, gg = rel - t * L.kg;
        const int ks = gg >> 1, part = gg & 1;  // part 0 = hi, 1 = lo
        const int n = 32 * t + (lane & 31), h = lane >> 5;
        const int ks0 = L.kg0 >> 1;             // k-steps of source 0
        const int src = ks < ks0 ? 0 : 1;
        const int kbase = 16 * (src ? ks - ks0 : ks) + 4 * h;
        for (int j = 0; j < 8; ++j) {
            const int k = kbase + (j & 3) + 8 * (j >> 2);
            float v = 0.f;
            if (k < L.kvalid[src]) {
                if (n < L.rows) v = L.w[(long)n * L.ld + L.col0[src] + k];
                else if (L.w_extra && n == L.extra_at && src == 0) v = L.w_extra[k];
            }
            unsigned bits;
            if (fmt == 0) {
                const unsigned hi = bf16_rne(v);
                bits = part == 0 ? hi : bf16_rne(v - __uint_as_float(hi << 16));
            } else {
                const unsigned hi = f16_rne(v);
                bits = part == 0 ? hi : f16_rne(v - f16_to_f32(hi));
            }
            w[j >> 1] |= bits << (16 * (j & 1));
        }
    }
    out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

// six-piece stream: fragment triple (3p .. 3p+2) = pieces p1, p2, p3 of one step (one n-tile of one 16-channel k-step;
// steps in K-major order within a layer); same lane / element order inside a fragment as the bf16x3 stream.  p1 = bf16(w), p2 = bf16(w - p1), p3 = bf16(w - p1 - p2).
__global__ void pack_bf16x6_kernel(PackDesc d, uint4* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= kX6StreamFrags * 64) return;
    const int f = gid >> 6, lane = gid & 63;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    const int part = f % kX6KFrags;
    if (f < kX6UsedFrags) {
        int l = 0;
        while (l + 1 < kNumLayers && f >= kX6KFrags * (d.L[l + 1].f0 / 2)) ++l;   // L.f0 counts two fragments per k-step
        const PackLayer& L = d.L[l];
        // step index within the layer, K-MAJOR (mlp_bf16x6.hip): all n-tiles of k-step 0, then of k-step 1, ...
        const int rel = (f - kX6KFrags * (L.f0 / 2)) / kX6KFrags;
        const int ks = rel / L.nt, t = rel - ks * L.nt;
        const int n = 32 * t + (lane & 31), h = lane >> 5;
        const int ks0 = L.kg0 >> 1;
        const int src = ks < ks0 ? 0 : 1;
        const int kbase = 16 * (src ? ks - ks0 : ks) + 4 * h;
        for (int j = 0; j < 8; ++j) {
            const int k = kbase + (j & 3) + 8 * (j >> 2);
            float v = 0.f;
            if (k < L.kvalid[src]) {
                if (n < L.rows) v = L.w[(long)n * L.ld + L.col0[src] + k];
                else if (L.w_extra && n == L.extra_at && src == 0) v = L.w_extra[k];
            }
            const unsigned p1 = bf16_rne(v);
            const float r1 = v - __uint_as_float(p1 << 16);
            const unsigned p2 = bf16_rne(r1);
            const float r2 = r1 - __uint_as_float(p2 << 16);
            const unsigned bits = part == 0 ? p1 : (part == 1 ? p2 : bf16_rne(r2));
            w[j >> 1] |= bits << (16 * (j & 1));
        }
    }
    out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

// plain-bf16 stream: one fragment per (tile, k-step): 8 bf16 per lane, same channel order
__global__ void pack_bf16_kernel(PackDesc d, uint4* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= kPlainStreamFrags * 64) return;
    const int f = gid >> 6, lane = gid & 63;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (f < kPlainUsedFrags) {
        int l = 0;
        while (l + 1 < kNumLayers && f >= d.L[l + 1].f0 / 2) ++l;
        const PackLayer& L = d.L[l];
        const int rel = f - L.f0 / 2, ksn = L.kg / 2;
        const int t = rel / ksn, ks = rel - t * ksn;
        const int n = 32 * t + (lane & 31), h = lane >> 5;
        const int ks0 = L.kg0 >> 1;
        const int src = ks < ks0 ? 0 : 1;
        const int kbase = 16 * (src ? ks - ks0 : ks) + 4 * h;
        for (int j = 0; j < 8; ++j) {
            const int k = kbase + (j & 3) + 8 * (j >> 2);
            float v = 0.f;
            if (k < L.kvalid[src]) {
                if (n < L.rows) v = L.w[(long)n * L.ld + L.col0[src] + k];
                else if (L.w_extra && n == L.extra_at && src == 0) v = L.w_extra[k];
            }
            w[j >> 1] |= bf16_rne(v) << (16 * (j & 1));
        }
    }
    out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

static void fill_pack_desc(const idn_facenerf_params& p, PackDesc& d);

int launch_pack_bf16(const idn_facenerf_params& p, float* packed, hipStream_t s) {
    PackDesc d;
    fill_pack_desc(p, d);
    const int total = kPlainStreamFrags * 64;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d, reinterpret_cast<uint4*>(packed));
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

int launch_pack_bf16x6(const idn_facenerf_params& p, float* packed, hipStream_t s) {
    PackDesc d;
    fill_pack_desc(p, d);
    const int total = kX6StreamFrags * 64;
    hipLaunchKernelGGL(pack_bf16x6_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d, reinterpret_cast<uint4*>(packed));
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

int launch_pack_bf16x3(const idn_facenerf_params& p, float* packed, hipStream_t s, int fmt) {
    PackDesc d;
    fill_pack_desc(p, d);
    const int total = kStreamFrags * 64;
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d, reinterpret_cast<uint4*>(packed), fmt);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

int launch_pack_f32(const idn_facenerf_params& p, float* packed, hipStream_t s) {
    PackDesc d;
    fill_pack_desc(p, d);
    const int total = kStreamFrags * 64;
    hipLaunchKernelGGL(pack_f32_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d,
                       reinterpret_cast<float4*>(packed));
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

static void fill_pack_desc(const idn_facenerf_params& p, PackDesc& d) {
    const int C = p.dim_aud + p.dim_expr + p.dim_latent;
    for (int l = 0; l < kNumLayers; ++l) {
        PackLayer& L = d.L[l];
        L.w_extra = nullptr;
        L.extra_at = -1;
        L.f0 = layer_f0(l);
        L.nt = kLayerNT[l];
        L.kg = kLayerKG[l];
        L.kg0 = L.kg;
        L.col0[0] = L.col0[1] = 0;
        L.kvalid[0] = L.kvalid[1] = 0;
        if (l == 0) {  // [PE(63) | cond(C)] -> 256 ; cond folded
            L.w = p.pts_w[0]; L.ld = IDN_PTS_CH + C; L.rows = IDN_W; L.kvalid[0] = IDN_PTS_CH;
        } else if (l == 5) {  // [PE(63) | cond(C) | h(256)] -> 256
            L.w = p.pts_w[5]; L.ld = IDN_PTS_CH + C + IDN_W; L.rows = IDN_W; L.kg0 = 8;
            L.kvalid[0] = IDN_PTS_CH; L.col0[1] = IDN_PTS_CH + C; L.kvalid[1] = IDN_W;
        } else if (l < 8) {
            L.w = p.pts_w[l]; L.ld = IDN_W; L.rows = IDN_W; L.kvalid[0] = IDN_W;
        } else if (l == 8) {  // [h(256) | dirPE(27) | expr] -> 128, plus alpha_linear as channel 128
            L.w = p.views_w[0]; L.ld = IDN_W + IDN_VIEWS_CH + p.dim_expr; L.rows = IDN_W / 2; L.kg0 = 32;
            L.kvalid[0] = IDN_W; L.col0[1] = IDN_W; L.kvalid[1] = IDN_VIEWS_CH;
            L.w_extra = p.alpha_w; L.extra_at = kSigmaChannel;
        } else if (l < 11) {
            L.w = p.views_w[l - 8]; L.ld = IDN_W / 2; L.rows = IDN_W / 2; L.kvalid[0] = IDN_W / 2;
        } else {
            L.w = p.rgb_w; L.ld = IDN_W / 2; L.rows = 3; L.kvalid[0] = IDN_W / 2;
        }
    }
}

struct FoldDesc {
    idn_facenerf_params p;
    const float* aud;
    const float* expr;
    const float* latent;
};

// cond[c] of the reference's `initial[:, 63:]` (face_nerf.py:45-55): aud | expr*1/3 | latent
__device__ __forceinline__ float cond_at(const FoldDesc& d, int c) {
    if (c < d.p.dim_aud) return d.aud[c];
    c -= d.p.dim_aud;
    if (c < d.p.dim_expr) return d.expr[c] * 1.0f / 3.0f;
    c -= d.p.dim_expr;
    return d.latent[c];
}

// One wavefront per output: the lanes split the conditioning columns of the row (a single thread per output
// walked them as one dependent fmaf chain: 67 us for 110 k MACs, four such launches per training step).
__global__ __launch_bounds__(256) void fold_kernel(FoldDesc d, float* out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
    if (o >= kBiasFloats) return;
    if (o >= kAlphaOff) {   // alpha_linear's / rgb_linear's weight rows, for the fp32 kernel's vector-unit heads
        if (lane == 0) out[o] = o < kRgbOff ? d.p.alpha_w[o - kAlphaOff] : d.p.rgb_w[o - kRgbOff];
        return;
    }
    int l = o / 256, n = o % 256;
    if (o >= bias_off(8)) {
        const int r = o - bias_off(8);
        if (r < 160) { l = 8; n = r; }
        else if (r < 160 + 128) { l = 9; n = r - 160; }
        else if (r < 160 + 256) { l = 10; n = r - 288; }
        else { l = 11; n = r - 416; }
    }
    const int C = d.p.dim_aud + d.p.dim_expr + d.p.dim_latent;
    float b = 0.f, dot = 0.f;
    if (l < 8) {
        b = d.p.pts_b[l][n];
        if (l == 0 || l == 5) {
            const int ld = IDN_PTS_CH + C + (l == 5 ? IDN_W : 0);
            const float* row = d.p.pts_w[l] + (long)n * ld + IDN_PTS_CH;
            for (int c = lane; c < C; c += 64) dot = fmaf(row[c], cond_at(d, c), dot);
        }
    } else if (l == 8) {
        if (n < IDN_W / 2) {
            b = d.p.views_b[0][n];
            if (d.expr) {
                const int ld = IDN_W + IDN_VIEWS_CH + d.p.dim_expr;
                const float* row = d.p.views_w[0] + (long)n * ld + IDN_W + IDN_VIEWS_CH;
                for (int e = lane; e < d.p.dim_expr; e += 64) dot = fmaf(row[e], d.expr[e] * 1.0f / 3.0f, dot);
            }
        } else if (n == kSigmaChannel) {
            b = d.p.alpha_b[0];
        }
    } else if (l < 11) {
        b = d.p.views_b[l - 8][n];
    } else if (n < 3) {
        b = d.p.rgb_b[n];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if (lane == 0) out[o] = b + dot;
}

int launch_fold(const idn_facenerf_params& p, const float* aud, const float* expr, const float* latent,
                float* folded, hipStream_t s) {
    FoldDesc d{p, aud, expr, latent};
    hipLaunchKernelGGL(fold_kernel, dim3((kBiasFloats + 3) / 4), dim3(256), 0, s, d, folded);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
