// Backward of one render pass (coarse or fine) for the training step
// (NeRFs/HeadNeRF/train/audio_exp_nerf.py:534-552: loss.backward() through raw2outputs,
// FaceNeRF and the per-frame conditioning).  gfx950, fp32 MFMA.
//
//   composite_bwd     d(rgb_map, rgb_fg, last_weight, acc) -> d raw           (one wave per ray)
//   delta chain       all delta_l = (delta_{l+1} . W_{l+1}) (.) [a_l > 0] in one fused kernel (mlp_f32_bwd.hip)
//   gemm_tn           dW_l = delta_l^T . a_{l-1}, contraction over points, split over workgroups
//   reduce_partials   sums the per-split dW blocks into the gradient tensors
//   (db_l = sum_p delta_l comes out of gemm_tn's pass over delta_l)
//   fold_bwd          conditioning columns of W0 / W5 / Wv0 and d aud, d latent
//
// Activations come from the training variant of the MLP kernel as row-major matrices
// (idn_internal.h, "activation slab").  Everything is deterministic: no float atomics.
#include "idn_internal.h"

namespace idn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int d_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------
// TN GEMM: part[split][n][k] = sum_{p in split} A[p][n0+n] * B[p][k0+k]
// Block = (64*NTW) x (64*KTW) outputs, waves 2 x 2, contraction chunk 32 points.
// ---------------------------------------------------------------------------
struct TNArgs {
    const float* A; int lda;   // delta  [P, >= N]
    const float* B; int ldb;   // acts   [P, >= K]
    float* part;               // [splits][N][K]
    int N, K;
    long P;                    // rows (multiple of 32)
    int chunks_per_split;      // 32-row chunks per split
    float* cpart;              // optional [splits][N]: column sums of A (the bias gradient), from the k-block-0 workgroups
};

// The two 32-row chunk tiles are double-buffered in LDS and filled by LDS-DMA (a chunk row is
// contiguous in global memory and in the tile, so one wave instruction moves 1 KiB of it): the
// loads of chunk c+1 are in flight while chunk c is multiplied, one barrier per chunk.  (Single
// buffered, every chunk paid its global-load latency: 70 % of the fp32 MFMA peak.)
template <int NTW, int KTW>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TNArgs g) {
    constexpr int BN = 64 * NTW, BK = 64 * KTW;
    constexpr int kTileFloats = 32 * (BN + BK);            // one chunk: A tile then B tile
    extern __shared__ __attribute__((aligned(16))) float tn_smem[];  // 2 * kTileFloats
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    const int n0 = blockIdx.x * BN, k0 = blockIdx.y * BK;
    const int split = blockIdx.z;
    f32x16 acc[NTW][KTW];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < KTW; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const long c_begin = (long)split * g.chunks_per_split;
    long c_end = c_begin + g.chunks_per_split;
    const long c_total = g.P / 32;
    if (c_end > c_total) c_end = c_total;
    const bool do_colsum = g.cpart != nullptr && blockIdx.y == 0 && tid < BN;
    float csum = 0.0f;

    // wave w moves every 4th 1-KiB piece of a tile: piece q holds float4 elements 64 q .. 64 q + 63
    auto fill = [&](long c, int buf) {
        const long p0 = c * 32;
        float* As = tn_smem + buf * kTileFloats;
        float* Bs = As + 32 * BN;
#pragma unroll
        for (int q = 0; q < BN / 32; ++q) {      // 32 * BN / 4 float4 = BN / 8 pieces, 4 waves -> BN / 32 each
            const int piece = 4 * q + w, e = piece * 64 + lane, row = e / (BN / 4), c4 = e % (BN / 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.A + (p0 + row) * g.lda + n0 + 4 * c4),
                                             (__attribute__((address_space(3))) void*)(As + piece * 256), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < BK / 32; ++q) {
            const int piece = 4 * q + w, e = piece * 64 + lane, row = e / (BK / 4), c4 = e % (BK / 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.B + (p0 + row) * g.ldb + k0 + 4 * c4),
                                             (__attribute__((address_space(3))) void*)(Bs + piece * 256), 16, 0, 0);
        }
    };
    if (c_begin < c_end) fill(c_begin, 0);
    for (long c = c_begin; c < c_end; ++c) {
        const int buf = (int)((c - c_begin) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk c have landed
        __syncthreads();                                   // everyone's have; everyone is done with the other buffer
        if (c + 1 < c_end) fill(c + 1, buf ^ 1);
        const float* As = tn_smem + buf * kTileFloats;
        const float* Bs = As + 32 * BN;
        if (do_colsum) {   // the delta tile is in LDS anyway: its column sums are the bias gradient (1 % more VALU)
#pragma unroll 8
            for (int r = 0; r < 32; ++r) csum += As[r * BN + tid];
        }
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int prow = 2 * s + hh;
            float a[NTW], b[KTW];
#pragma unroll
            for (int x = 0; x < NTW; ++x) a[x] = As[prow * BN + 32 * NTW * wr + 32 * x + i];
#pragma unroll
            for (int y = 0; y < KTW; ++y) b[y] = Bs[prow * BK + 32 * KTW * wc + 32 * y + i];
#pragma unroll
            for (int x = 0; x < NTW; ++x)
#pragma unroll
                for (int y = 0; y < KTW; ++y) acc[x][y] = mfma32(a[x], b[y], acc[x][y]);
        }
    }
    if (do_colsum) g.cpart[(long)split * g.N + n0 + tid] = csum;
    float* out = g.part + (long)split * g.N * g.K;
#pragma unroll
    for (int x = 0; x < NTW; ++x)
#pragma unroll
        for (int y = 0; y < KTW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 32 * NTW * wr + 32 * x + d_row(r, hh);
                const int k = k0 + 32 * KTW * wc + 32 * y + i;
                out[(long)n * g.K + k] = acc[x][y][r];
            }
}

// out[n*ldo + k] = sum_s part[s][n][k],  n < rows, k < cols  (rows/cols may be smaller than N/K: padding dropped)
// 64 outputs x 4 split lanes per block: lane q adds splits q, q+4, ... in fp64, the four lanes are
// then added in a fixed order (deterministic), so 4x as many loads are in flight per output.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int splits, int N, int K,
                                                              float* __restrict__ out, int ldo, int rows, int cols) {
    __shared__ double red[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;
    const bool live = idx < rows * cols;
    const int n = live ? idx / cols : 0, k = live ? idx % cols : 0;
    double s = 0.0;
    if (live) {
        const float* src = part + (long)n * K + k;
        const long stride = (long)N * K;
#pragma unroll 8
        for (int sp = q; sp < splits; sp += 4) s += (double)src[sp * stride];
    }
    red[q][o] = s;
    __syncthreads();
    if (q == 0 && live) out[(long)n * ldo + k] = (float)(((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]);
}

// ---------------------------------------------------------------------------
// compositing backward (baseline.py:325-375 differentiated; cumprod's gradient as
// PyTorch computes it when no factor is zero: reverse cumsum(grad * out) / input).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double shfl_up_dd(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_dd(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

struct CompBwdArgs {
    const float4* raw; const float* z; const float* rays; const float* bc;
    const float* g_rgb;  // [n,3] d rgb_map (may be null)
    const float* g_fg;   // [n,3] d rgb_fg  (may be null)
    const float* g_lw;   // [n]   d last_weight (may be null)
    const float* g_acc;  // [n]   d acc_map (may be null)
    float* d_rgb; int ld_rgb;    // row p = ray*S+s: d raw_rgb at d_rgb[p*ld_rgb + 0..2]
    float* d_sig; int ld_sig;    // d raw_sigma at d_sig[p*ld_sig]
    long n_rays; int S;
};

template <int SPL>
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.n_rays) return;
    const int S = a.S;
    const float* rr = a.rays + ray * IDN_RAY_FLOATS;
    const float dn = sqrtf((rr[3] * rr[3] + rr[4] * rr[4]) + rr[5] * rr[5]);
    const float4* rawr = a.raw + ray * S;
    const float* zr = a.z + ray * S;
    float gr = 0, gg = 0, gb = 0, fr = 0, fg = 0, fb = 0, glw = 0, gacc = 0;
    if (a.g_rgb) { gr = a.g_rgb[ray * 3]; gg = a.g_rgb[ray * 3 + 1]; gb = a.g_rgb[ray * 3 + 2]; }
    if (a.g_fg) { fr = a.g_fg[ray * 3]; fg = a.g_fg[ray * 3 + 1]; fb = a.g_fg[ray * 3 + 2]; }
    if (a.g_lw) glw = a.g_lw[ray];
    if (a.g_acc) gacc = a.g_acc[ray];

    float zs[SPL + 1];
    float4 rw[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        zs[i] = s < S ? zr[s] : 0.f;
        rw[i] = s < S ? rawr[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    zs[SPL] = __shfl_down(zs[0], 1, 64);
    float alpha[SPL], tf[SPL], ex[SPL], dist[SPL];
    double local = 1.0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        float d = (s >= S - 1) ? 1e10f : (zs[i + 1] - zs[i]);
        d = d * dn;
        const float e = expf(-(fmaxf(rw[i].w, 0.0f) + 1e-6f) * d);
        dist[i] = d;
        ex[i] = e;
        alpha[i] = (s < S) ? 1.0f - e : 0.0f;
        tf[i] = (s < S) ? (1.0f - (1.0f - e)) + 1e-10f : 1.0f;
        local *= (double)tf[i];
    }
    // exclusive prefix product over lanes
    double incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_dd(incl, d);
        if (lane >= d) incl *= o;
    }
    double run = shfl_up_dd(incl, 1);
    if (lane == 0) run = 1.0;
    float T[SPL], w[SPL], dw[SPL], cr[SPL], cg[SPL], cb[SPL];
    double lsum = 0.0;  // sum of dw*w over this lane's samples
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        T[i] = (float)run;
        w[i] = alpha[i] * T[i];
        run *= (double)tf[i];
        dw[i] = 0.f;
        cr[i] = cg[i] = cb[i] = 0.f;
        if (s < S) {
            if (s == S - 1) {
                cr[i] = a.bc[ray * 3]; cg[i] = a.bc[ray * 3 + 1]; cb[i] = a.bc[ray * 3 + 2];
                dw[i] = (gr * cr[i] + gg * cg[i] + gb * cb[i]) + glw + gacc;
            } else {
                cr[i] = 1.0f / (1.0f + expf(-rw[i].x));
                cg[i] = 1.0f / (1.0f + expf(-rw[i].y));
                cb[i] = 1.0f / (1.0f + expf(-rw[i].z));
                dw[i] = ((gr + fr) * cr[i] + (gg + fg) * cg[i] + (gb + fb) * cb[i]) + gacc;
            }
            lsum += (double)dw[i] * (double)w[i];
        }
    }
    // suffix sums of dw*w: total - inclusive prefix
    double pin = lsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_dd(pin, d);
        if (lane >= d) pin += o;
    }
    double total = lsum;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) total += shfl_xor_dd(total, m);
    double before = pin - lsum;  // sum over earlier lanes
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        if (s < S) {
            before += (double)dw[i] * (double)w[i];          // inclusive up to s
            const double suffix = total - before;            // sum_{t > s} dw_t w_t
            const float dalpha = dw[i] * T[i] - (float)(suffix / (double)tf[i]);
            const float dsig = (rw[i].w > 0.f) ? dalpha * ex[i] * dist[i] : 0.f;
            const long p = ray * S + s;
            a.d_sig[p * a.ld_sig] = dsig;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            if (s < S - 1) {
                d0 = w[i] * (gr + fr) * cr[i] * (1.0f - cr[i]);
                d1 = w[i] * (gg + fg) * cg[i] * (1.0f - cg[i]);
                d2 = w[i] * (gb + fb) * cb[i] * (1.0f - cb[i]);
            }
            a.d_rgb[p * a.ld_rgb + 0] = d0;
            a.d_rgb[p * a.ld_rgb + 1] = d1;
            a.d_rgb[p * a.ld_rgb + 2] = d2;
        }
    }
}

// ---------------------------------------------------------------------------
// conditioning fold backward
//   dW0[:, 63+c] = db0'[n] cond[c];  dW5[:, 63+c] = db5'[n] cond[c];  dWv0[:, 283+e] = dbv'[n] expr3[e]
//   d cond[c] = sum_n W0[n][63+c] db0'[n] + W5[n][63+c] db5'[n]   -> d aud, d latent (accumulated)
// ---------------------------------------------------------------------------
struct FoldBwdArgs {
    idn_facenerf_params p;
    const float* aud; const float* expr; const float* latent;
    const float* db0; const float* db5; const float* dbv;  // [256], [256], [128]
    float* gW0; float* gW5; float* gWv0;                    // gradient tensors (full nn.Linear layout)
    float* d_aud; float* d_latent;                          // accumulated (+=), may be null
};
__device__ __forceinline__ float cond_val(const FoldBwdArgs& d, int c) {
    if (c < d.p.dim_aud) return d.aud[c];
    c -= d.p.dim_aud;
    if (c < d.p.dim_expr) return d.expr[c] * 1.0f / 3.0f;
    c -= d.p.dim_expr;
    return d.latent[c];
}
__global__ void fold_bwd_kernel(FoldBwdArgs d) {
    const int C = d.p.dim_aud + d.p.dim_expr + d.p.dim_latent;
    const int ld0 = IDN_PTS_CH + C, ld5 = IDN_PTS_CH + C + IDN_W, ldv = IDN_W + IDN_VIEWS_CH + d.p.dim_expr;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_outer = IDN_W * C;
    if (idx < n_outer) {
        const int n = idx / C, c = idx % C;
        const float cv = cond_val(d, c);
        d.gW0[(long)n * ld0 + IDN_PTS_CH + c] = d.db0[n] * cv;
        d.gW5[(long)n * ld5 + IDN_PTS_CH + c] = d.db5[n] * cv;
    } else if (idx < n_outer + (IDN_W / 2) * d.p.dim_expr) {
        const int k = idx - n_outer, n = k / d.p.dim_expr, e = k % d.p.dim_expr;
        d.gWv0[(long)n * ldv + IDN_W + IDN_VIEWS_CH + e] = d.dbv[n] * (d.expr[e] * 1.0f / 3.0f);
    } else {
        const int c = idx - n_outer - (IDN_W / 2) * d.p.dim_expr;
        if (c >= C) return;
        double s = 0.0;
        for (int n = 0; n < IDN_W; ++n)
            s += (double)d.p.pts_w[0][(long)n * ld0 + IDN_PTS_CH + c] * (double)d.db0[n] +
                 (double)d.p.pts_w[5][(long)n * ld5 + IDN_PTS_CH + c] * (double)d.db5[n];
        if (c < d.p.dim_aud) {
            if (d.d_aud) d.d_aud[c] += (float)s;
        } else if (c >= d.p.dim_aud + d.p.dim_expr) {
            if (d.d_latent) d.d_latent[c - d.p.dim_aud - d.p.dim_expr] += (float)s;
        }
    }
}

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr int kMaxSplits = 256;
constexpr int kColsumBlocks = 256;   // rows of the column-sum partial buffer (>= kMaxSplits)

struct BwdWs {
    float *dA[8], *dV[2], *dV0, *dRGB, *part, *cpart, *dbtmp, *wbwd;
    size_t bytes;
};
static BwdWs carve_bwd(char* base, int64_t p_pad) {
    BwdWs w;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = reinterpret_cast<float*>(base + off);
        off += al256(floats * 4);
        return p;
    };
    for (int l = 0; l < 8; ++l) w.dA[l] = take((size_t)p_pad * 256);  // delta of pts_linears.l (pre-activation)
    w.dV[0] = take((size_t)p_pad * 128);
    w.dV[1] = take((size_t)p_pad * 128);
    w.dV0 = take((size_t)p_pad * 256);
    w.dRGB = take((size_t)p_pad * 64);
    w.part = take((size_t)kMaxSplits * 256 * 256);
    w.cpart = take((size_t)kColsumBlocks * 256);
    w.dbtmp = take(1024);
    w.wbwd = take((size_t)kBwdStreamFrags * kFragFloats);               // transposed weight stream of the delta chain
    w.bytes = off;
    return w;
}

size_t bwd_workspace_bytes(int64_t n_points) {
    const int64_t p_pad = (n_points + 127) / 128 * 128;
    return carve_bwd(nullptr, p_pad).bytes;
}

// part[split][N][K] = A[:, :N]^T . B[:, :K] over point splits; returns the split count
static int run_tn_partials(const float* A, int lda, int N, const float* B, int ldb, int K, int64_t P, float* part,
                           int* splits_out, hipStream_t s, float* cpart = nullptr) {
    int ntw, ktw;
    if (N == 256 && K == 256) { ntw = 4; ktw = 4; }
    else if (N == 256 && K == 64) { ntw = 4; ktw = 1; }
    else if (N == 128 && K == 128) { ntw = 2; ktw = 2; }
    else if (N == 64 && K == 128) { ntw = 1; ktw = 2; }
    else return fail(IDN_EUNSUPPORTED, "gemm_tn: no instantiation for %d x %d", N, K);
    const int bx = N / (64 * ntw), by = K / (64 * ktw);
    const long chunks = P / 32;
    int splits = kMaxSplits / (bx * by);
    if (splits > chunks) splits = (int)chunks;
    if (splits < 1) splits = 1;
    const int cps = (int)((chunks + splits - 1) / splits);
    splits = (int)((chunks + cps - 1) / cps);
    TNArgs g{A, lda, B, ldb, part, N, K, (long)P, cps, cpart};
    const dim3 grid(bx, by, splits), block(256);
    const size_t lds = 2 * 32 * (size_t)(64 * ntw + 64 * ktw) * 4;
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<4, 4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 512 * 4));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<4, 1>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 320 * 4));
            return IDN_OK;
        }, &num_cu))
        return e;
    ProfScope prof(s, P, IDN_PROF_DW_GEMM);
    if (ntw == 4 && ktw == 4) hipLaunchKernelGGL((gemm_tn_kernel<4, 4>), grid, block, lds, s, g);
    else if (ntw == 4 && ktw == 1) hipLaunchKernelGGL((gemm_tn_kernel<4, 1>), grid, block, lds, s, g);
    else if (ntw == 2 && ktw == 2) hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), grid, block, lds, s, g);
    else hipLaunchKernelGGL((gemm_tn_kernel<1, 2>), grid, block, lds, s, g);
    IDN_HIP_CHECK(hipGetLastError());
    *splits_out = splits;
    return IDN_OK;
}
// out[(0..rows) x (0..cols)] (ld ldo) = sum over splits of part rows row0.. of the N x K product
static int run_reduce(const float* part, int splits, int N, int K, int row0, float* out, int ldo, int rows, int cols,
                      hipStream_t s) {
    const int total = rows * cols;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((total + 63) / 64), dim3(256), 0, s, part + (size_t)row0 * K,
                       splits, N, K, out, ldo, rows, cols);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}
// out = A^T B (rows x cols of it); db (optional, `db_cols` entries) = column sums of A, from the same pass over A
static int run_tn(const float* A, int lda, int N, const float* B, int ldb, int K, int64_t P, float* part, float* out,
                  int ldo, int rows, int cols, hipStream_t s, float* cpart = nullptr, float* db = nullptr, int db_cols = 0) {
    int splits = 0;
    if (int e = run_tn_partials(A, lda, N, B, ldb, K, P, part, &splits, s, db ? cpart : nullptr)) return e;
    if (db)
        if (int e = run_reduce(cpart, splits, 1, N, 0, db, N, 1, db_cols, s)) return e;
    return run_reduce(part, splits, N, K, 0, out, ldo, rows, cols, s);
}

int launch_pass_bwd(const idn_facenerf_params& p, const idn_facenerf_grads& gr, const float* aud, const float* expr,
                    const float* latent, const float* acts, const float* raw, const float* z, const float* rays,
                    const float* bc, int64_t n_rays, int S, const float* g_rgb, const float* g_fg, const float* g_lw,
                    const float* g_acc, float* d_aud, float* d_latent, void* ws_, size_t ws_bytes, hipStream_t s) {
    const int64_t P = n_rays * S;
    const int64_t Pp = (P + 127) / 128 * 128;
    if (S < 2 || S > 256) return fail(IDN_EUNSUPPORTED, "pass_bwd: n_samples %d outside [2, 256]", S);
    const BwdWs w = carve_bwd(reinterpret_cast<char*>(ws_), Pp);
    if (!ws_ || ws_bytes < w.bytes) return fail(IDN_EWORKSPACE, "backward workspace %zu < %zu", ws_bytes, w.bytes);
    const int C = p.dim_aud + p.dim_expr + p.dim_latent;
    const int ld0 = IDN_PTS_CH + C, ld5 = IDN_PTS_CH + C + IDN_W, ldv = IDN_W + IDN_VIEWS_CH + p.dim_expr;
    auto act = [&](int i) { return acts + (size_t)act_off(i) * Pp; };
    auto a_l = [&](int l) { return act(kActA1 + l - 1); };  // post-ReLU output of pts_linears.(l-1), l = 1..8
    auto v_l = [&](int l) { return act(kActV1 + l - 1); };  // post-ReLU output of views_linears.(l-1), l = 1..3

    // d(outputs) -> d raw, written straight into the head deltas (zero elsewhere)
    IDN_HIP_CHECK(hipMemsetAsync(w.dRGB, 0, (size_t)Pp * 64 * 4, s));
    IDN_HIP_CHECK(hipMemsetAsync(w.dV0, 0, (size_t)Pp * 256 * 4, s));
    {
        CompBwdArgs a{reinterpret_cast<const float4*>(raw), z, rays, bc, g_rgb, g_fg, g_lw, g_acc,
                      w.dRGB, 64, w.dV0 + kSigmaChannel, 256, (long)n_rays, S};
        const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
        switch ((S + 63) / 64) {
            case 1: hipLaunchKernelGGL(composite_bwd_kernel<1>, grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL(composite_bwd_kernel<2>, grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL(composite_bwd_kernel<3>, grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL(composite_bwd_kernel<4>, grid, block, 0, s, a); break;
        }
        IDN_HIP_CHECK(hipGetLastError());
    }
#define TRY(x) do { if (int e_ = (x)) return e_; } while (0)
    // All pre-activation deltas in one fused pass over the points (mlp_f32_bwd.hip): dV[0] = delta of
    // views_linears.2, dV[1] = views_linears.1, dV0[:, :128] = views_linears.0 (col 128 = d sigma), dA[l] = pts_linears.l
    TRY(launch_pack_f32_bwd(p, w.wbwd, s));
    TRY(launch_delta_chain(w.wbwd, acts, Pp, w.dRGB, w.dV0, w.dV[0], w.dV[1], w.dA, s));
    // weight and bias gradients: dW_l = delta_l^T a_{l-1} (contraction over the points), db_l = column sums
    TRY(run_tn(w.dRGB, 64, 64, v_l(3), 128, 128, Pp, w.part, gr.rgb_w, 128, 3, 128, s, w.cpart, gr.rgb_b, 3));
    TRY(run_tn(w.dV[0], 128, 128, v_l(2), 128, 128, Pp, w.part, gr.views_w[2], 128, 128, 128, s, w.cpart, gr.views_b[2], 128));
    TRY(run_tn(w.dV[1], 128, 128, v_l(1), 128, 128, Pp, w.part, gr.views_w[1], 128, 128, 128, s, w.cpart, gr.views_b[1], 128));
    // views_linears.0 (+ alpha_linear as channel 128); inputs [a8 | dirPE | expr(folded)]
    {
        int splits = 0;
        TRY(run_tn_partials(w.dV0, 256, 256, a_l(8), 256, 256, Pp, w.part, &splits, s, w.cpart));
        TRY(run_reduce(w.cpart, splits, 1, 256, 0, w.dbtmp, 256, 1, 256, s));
        TRY(run_reduce(w.part, splits, 256, 256, 0, gr.views_w[0], ldv, 128, 256, s));
        TRY(run_reduce(w.part, splits, 256, 256, kSigmaChannel, gr.alpha_w, 256, 1, 256, s));
        TRY(run_tn_partials(w.dV0, 256, 256, act(kActDir), 64, 64, Pp, w.part, &splits, s));
        TRY(run_reduce(w.part, splits, 256, 64, 0, gr.views_w[0] + IDN_W, ldv, 128, IDN_VIEWS_CH, s));
        IDN_HIP_CHECK(hipMemcpyAsync(gr.views_b[0], w.dbtmp, 128 * 4, hipMemcpyDeviceToDevice, s));
        IDN_HIP_CHECK(hipMemcpyAsync(gr.alpha_b, w.dbtmp + kSigmaChannel, 4, hipMemcpyDeviceToDevice, s));
    }
    for (int l = 7; l >= 1; --l) {
        const float* cur = w.dA[l];
        if (l == 5) {
            TRY(run_tn(cur, 256, 256, a_l(5), 256, 256, Pp, w.part, gr.pts_w[5] + IDN_PTS_CH + C, ld5, 256, 256, s, w.cpart, gr.pts_b[5], 256));
            TRY(run_tn(cur, 256, 256, act(kActX0), 64, 64, Pp, w.part, gr.pts_w[5], ld5, 256, IDN_PTS_CH, s));
        } else {
            TRY(run_tn(cur, 256, 256, a_l(l), 256, 256, Pp, w.part, gr.pts_w[l], 256, 256, 256, s, w.cpart, gr.pts_b[l], 256));
        }
    }
    TRY(run_tn(w.dA[0], 256, 256, act(kActX0), 64, 64, Pp, w.part, gr.pts_w[0], ld0, 256, IDN_PTS_CH, s, w.cpart, gr.pts_b[0], 256));
#undef TRY
    {
        FoldBwdArgs f{p, aud, expr, latent, gr.pts_b[0], gr.pts_b[5], gr.views_b[0], gr.pts_w[0], gr.pts_w[5],
                      gr.views_w[0], d_aud, d_latent};
        const int total = IDN_W * C + (IDN_W / 2) * p.dim_expr + C;
        if (total > 0) {
            hipLaunchKernelGGL(fold_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, f);
            IDN_HIP_CHECK(hipGetLastError());
        }
    }
    return IDN_OK;
}

}  // namespace idn
