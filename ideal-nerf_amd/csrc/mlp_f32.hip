// Fused positional-encoding + FaceNeRF MLP forward, fp32 MFMA (gfx950).
//
// Replaces, per point: Embedder.embed (NeRFs/HeadNeRF/helper.py:174-224),
// Network.run_network (NeRFs/HeadNeRF/train/audio_exp_nerf.py:376-394) and
// FaceNeRF.forward (models/face_nerf.py:40-80).
//
// Shape of the computation (see DESIGN.md "MLP kernel"):
//   * one wave = 32 points; it computes H^T = W . X^T with v_mfma_f32_32x32x2_f32, weights
//     as the A operand (streamed), activations as the B operand (registers).  The
//     accumulator layout of one layer IS the B-operand layout of the next, so the 256-wide
//     activation of every layer lives in 128 VGPRs and never touches LDS or HBM;
//   * a workgroup = 4 waves (one per SIMD, 512-register budget) = 128 points sharing one
//     weight stream: global -> LDS by global_load_lds_dwordx4 into a 2 x 64 KiB ring, one
//     barrier per 64 KiB slice (= 256 MFMAs per wave);
//   * workgroups are persistent (one per CU) and walk the point tiles grid-stride, so the
//     weight stream keeps flowing across tiles.
#include "idn_internal.h"
#include <utility>

namespace idn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
using ic = std::integral_constant<int, N>;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Diagnostic build only (-DIDN_DIAG): per-wave cycle totals by category, written to a buffer
// nothing else reads.  Never compiled into the shipped library (cdna_hip_programming.md 7).
#ifdef IDN_DIAG
#define DIAG_ONLY(x) x
__device__ unsigned long long g_diag[8];
enum { kDgTotal = 0, kDgInput = 1, kDgBarrier = 2, kDgBoundary = 3, kDgStore = 4 };
struct Diag {
    unsigned long long acc[5] = {0, 0, 0, 0, 0};
    unsigned long long t0 = 0;
    __device__ __forceinline__ void begin() { __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void end(int cat) {
        __builtin_amdgcn_sched_barrier(0);
        acc[cat] += clock64() - t0;
        __builtin_amdgcn_sched_barrier(0);
    }
};
#define DIAG_BEGIN(d) (d).begin()
#define DIAG_END(d, c) (d).end(c)
#else
#define DIAG_ONLY(x)
struct Diag {};
#define DIAG_BEGIN(d)
#define DIAG_END(d, c)
#endif

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------
// weight stream: global -> LDS ring
//
// The ring holds two 64 KiB slices.  A slice is fetched by 16 "pieces" (one
// global_load_lds_dwordx4 per wave each = 4 KiB per piece); the pieces of slice s+1 are
// issued one per fragment-pair step during the first half of slice s, so their address
// arithmetic rides in MFMA shadows instead of stalling the restart after a barrier.
// ---------------------------------------------------------------------------
constexpr int kPieces = kSliceFrags / 4;  // 16

struct WStream {
    Diag* dg;
    const char* gbase;  // stream start + this lane's 16-byte column
    const char* gnext;  // same, for the slice currently being fetched
    int next_slice;
    char* ring_wave;    // ring + wave * 1 KiB (wave-uniform LDS destination base)

    __device__ __forceinline__ void advance() {
        gnext += kSliceBytes;
        if (++next_slice == kNumSlices) {
            next_slice = 0;
            gnext = gbase;
        }
    }
    template <int SLOT, int J>
    __device__ __forceinline__ void issue_piece() {
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gnext + J * 4096),
                                         LDS_PTR(ring_wave + SLOT * kSliceBytes + J * 4096), 16, 0, 0);
        if constexpr (J == kPieces - 1) advance();
    }
    template <int SLOT, int J0>
    __device__ __forceinline__ void issue_rest() {
        static_for<kPieces - J0>([&](auto I) { issue_piece<SLOT, J0 + decltype(I)::value>(); });
    }
    // every wave's share of the slice to be read next has landed (vmcnt(0) precedes the
    // barrier) and every wave is done reading the other slot
    __device__ __forceinline__ void open_slice() {
        DIAG_BEGIN(*dg);
        __syncthreads();
        DIAG_END(*dg, kDgBarrier);
    }
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// A-fragment reads.  hipcc (ROCm 7.2) waits lgkmcnt(0) after a prefetching ds_read -- i.e.
// for the read it has just issued.  The reads are therefore issued from inline asm, which
// the compiler does not count, and retired by a counted wait tied to the destination
// registers ("+v"): LDS operations return in order, so lgkmcnt(2) right after issuing pair
// p+1 means pair p has landed.  Compiler-issued LDS/SMEM operations in between only make
// these waits stricter (never weaker), and its own counted waits likewise
// (cdna_hip_programming.md section 5.7).
// ---------------------------------------------------------------------------
struct FragReader {
    uint32_t addr0, addr1;  // LDS byte address of this lane's 16 bytes in fragment 0 / fragment 64
    f32x4 pref0, pref1;     // fragment pair issued ahead of its consumer (layer / slice start)

    template <int F>
    __device__ __forceinline__ f32x4 issue() const {
        constexpr int fr = F % kRingFrags;
        f32x4 v;
        if constexpr (fr < 64)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr0), "n"(fr * kFragBytes) : "memory");
        else
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr1), "n"((fr - 64) * kFragBytes) : "memory");
        return v;
    }
    // all but the newest `Newer` LDS reads of this wave have completed => v0, v1 are valid
    template <int Newer>
    static __device__ __forceinline__ void retire(f32x4& v0, f32x4& v1) {
        if constexpr (Newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v0), "+v"(v1)::"memory");
    }
};

// One layer: out[t] += W[32t.., :] . B  for t = 0..NT-1, tile after tile (the stream is
// tile-major), each tile's KG k-groups consumed as KG/2 fragment pairs = 8 MFMAs per step.
//   * the reads of pair p+1 are issued ahead of the MFMAs of pair p;
//   * one prefetch piece of the next slice is issued per step in the first half of a slice;
//   * side(t, s, half) is called after each group of four MFMAs: VALU / LDS work placed
//     there issues while the matrix pipe is busy (finished tiles' ReLU, the next tile's bias).
// F0 = index of the layer's first fragment in the stream.  On entry fr.pref* hold pair F0
// in flight unless F0 opens a slice; on exit they hold pair F0 + NT*KG likewise.
template <int F0, int NT, int KG, class BGet, class Side>
__device__ __forceinline__ void run_layer(f32x16 (&out)[NT], BGet&& bget, WStream& ws, FragReader& fr, Side&& side) {
    constexpr int STEPS = KG / 2, NP = NT * STEPS;
    static_assert(KG % 2 == 0 && F0 % 2 == 0, "fragments are consumed in pairs");
    if constexpr (F0 % kSliceFrags == 0) {
        ws.open_slice();
        fr.pref0 = fr.template issue<F0>();
        fr.pref1 = fr.template issue<F0 + 1>();
    }
    f32x4 a0 = fr.pref0, a1 = fr.pref1;
    static_for<NP>([&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        constexpr int t = pi / STEPS, s = pi % STEPS, g0 = 2 * s, g1 = g0 + 1;
        constexpr int f = F0 + 2 * pi;
        constexpr bool next_crosses = ((f + 2) % kSliceFrags == 0);
        constexpr int jpos = (f % kSliceFrags) / 2;   // position of this pair inside its slice
        constexpr int slot = (f / kSliceFrags) & 1;
        f32x4 n0 = a0, n1 = a1;
        if constexpr (!next_crosses) {
            n0 = fr.template issue<f + 2>();
            n1 = fr.template issue<f + 3>();
            FragReader::retire<2>(a0, a1);
        } else {
            FragReader::retire<0>(a0, a1);
        }
        if constexpr (jpos < kPieces) ws.template issue_piece<slot ^ 1, jpos>();
        out[t] = mfma(a0.x, bget(ic<g0>{}, ic<0>{}), out[t]);
        out[t] = mfma(a0.y, bget(ic<g0>{}, ic<1>{}), out[t]);
        out[t] = mfma(a0.z, bget(ic<g0>{}, ic<2>{}), out[t]);
        out[t] = mfma(a0.w, bget(ic<g0>{}, ic<3>{}), out[t]);
        side(ic<t>{}, ic<s>{}, ic<0>{});
        out[t] = mfma(a1.x, bget(ic<g1>{}, ic<0>{}), out[t]);
        out[t] = mfma(a1.y, bget(ic<g1>{}, ic<1>{}), out[t]);
        out[t] = mfma(a1.z, bget(ic<g1>{}, ic<2>{}), out[t]);
        out[t] = mfma(a1.w, bget(ic<g1>{}, ic<3>{}), out[t]);
        side(ic<t>{}, ic<s>{}, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (next_crosses && pi + 1 < NP) {
            ws.open_slice();
            n0 = fr.template issue<f + 2>();
            n1 = fr.template issue<f + 3>();
        }
        a0 = n0;
        a1 = n1;
    });
    fr.pref0 = a0;  // pair F0 + NT*KG (already in flight) when it does not open a slice
    fr.pref1 = a1;
}

// End of a pass: the pieces of the next pass's first slice that the (short) last slice did
// not get to issue.
template <int F_END>
__device__ __forceinline__ void finish_pass(WStream& ws) {
    static_assert(F_END % kSliceFrags != 0 && (F_END + kSliceFrags - 1) / kSliceFrags == kNumSlices,
                  "the last consumed slice must be the stream's last and partially used");
    constexpr int jpos = (F_END % kSliceFrags) / 2;
    constexpr int slot = (F_END / kSliceFrags) & 1;
    if constexpr (jpos < kPieces) ws.template issue_rest<slot ^ 1, jpos>();
}

// acc[4q..4q+3] of one tile <- bias of channels 32t + 8q + 4h + 0..3
template <int Q>
__device__ __forceinline__ void bias_quad(f32x16& tile, const float* bias_tile_half /* bias_s + off + 32t + 4h */) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_tile_half + 8 * Q);
    tile[4 * Q + 0] = b.x;
    tile[4 * Q + 1] = b.y;
    tile[4 * Q + 2] = b.z;
    tile[4 * Q + 3] = b.w;
}
__device__ __forceinline__ void bias_tile(f32x16& tile, const float* bias_tile_half) {
    bias_quad<0>(tile, bias_tile_half);
    bias_quad<1>(tile, bias_tile_half);
    bias_quad<2>(tile, bias_tile_half);
    bias_quad<3>(tile, bias_tile_half);
}
template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NT], const float* bias_half /* bias_s + off + 4h */) {
    static_for<NT>([&](auto T) { bias_tile(acc[decltype(T)::value], bias_half + 32 * decltype(T)::value); });
}

__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }  // one v_med3_f32
// ReLU, in place, of registers [R0, R0 + CNT) of a tile (clipped to the 16 a tile has)
template <int R0, int CNT>
__device__ __forceinline__ void relu_regs(f32x16& tile) {
    static_for<CNT>([&](auto R) {
        constexpr int r = R0 + decltype(R)::value;
        if constexpr (r < 16) tile[r] = relu1(tile[r]);
    });
}
template <int NT>
__device__ __forceinline__ void relu_tiles(f32x16 (&t)[NT]) {
    static_for<NT>([&](auto T) { relu_regs<0, 16>(t[decltype(T)::value]); });
}

// The in-shadow work of one layer.
//   * tile t-1 (finished) is ReLU'd while tile t accumulates;
//   * the previous layer's last tile (`deferred`, when DEFER) is ReLU'd during the first half
//     of tile 0 -- before any k-group that reads it (it is the LAST tile of the input);
//   * the bias of tile t+1 is loaded during the first four steps of tile t.
// The last tile's own ReLU is left to the next layer's `deferred`.
template <int NT, int STEPS, bool DEFER>
struct LayerSide {
    f32x16* out;
    f32x16* deferred;
    const float* bias_half;  // bias_s + layer offset + 4h
    template <int T, int S, int H>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>, ic<H>) const {
        constexpr int C = (16 + STEPS - 1) / STEPS;          // ReLU registers per step
        constexpr int CD = (32 + STEPS - 1) / STEPS;         // deferred tile: done by STEPS/2
        if constexpr (H == 0) {
            if constexpr (T > 0) relu_regs<S * C, C>(out[T - 1]);
            if constexpr (T == 0 && DEFER) relu_regs<S * CD, CD>(*deferred);
        } else {
            if constexpr (T + 1 < NT && S < 4) bias_quad<S>(out[T + 1], bias_half + 32 * (T + 1));
        }
    }
};
struct NoSide {
    template <int T, int S, int H>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>, ic<H>) const {}
};

// sin / cos of 2*pi*r for |r| <= 1/8 (Taylor in r; the first dropped terms are < 2e-9).
__device__ __forceinline__ void sincos_2pi_small(float r, float& sn, float& cs) {
    const float s = r * r;
    float ps = 4.2058693945e+01f;                 //  (2pi)^9 / 9!
    ps = fmaf(ps, s, -7.6705859753e+01f);         // -(2pi)^7 / 7!
    ps = fmaf(ps, s, 8.1605249276e+01f);          //  (2pi)^5 / 5!
    ps = fmaf(ps, s, -4.1341702240e+01f);         // -(2pi)^3 / 3!
    ps = fmaf(ps, s, 6.2831853072e+00f);          //   2pi
    sn = ps * r;
    float pc = -2.6426256783e+01f;                // -(2pi)^10 / 10!
    pc = fmaf(pc, s, 6.0244641371e+01f);          //  (2pi)^8 / 8!
    pc = fmaf(pc, s, -8.5456817206e+01f);         // -(2pi)^6 / 6!
    pc = fmaf(pc, s, 6.4939394023e+01f);          //  (2pi)^4 / 4!
    pc = fmaf(pc, s, -1.9739208802e+01f);         // -(2pi)^2 / 2!
    cs = fmaf(pc, s, 1.0f);
}

// gamma_L(v) for a 3-vector, as 3 + 6L features in the reference's order
// (helper.py:183-201): [v, sin(2^0 v), cos(2^0 v), ..., sin(2^(L-1) v), cos(2^(L-1) v)],
// zero padded to NF.
//
// The reference evaluates sin/cos of fl32(2^b v), and 2^b v is exact, so the true
// argument is known exactly: reduce the phase p = v / 2pi once per axis in fp64, double it
// per band (exact), split off the quadrant (exact) and evaluate a short polynomial on
// |r| <= 1/8 of a turn.  Measured against torch.sin/cos on CPU: max |diff| 1.2e-7 on every
// band (<= 2 ulp at 1.0), with no data-dependent branch (ocml's sincosf takes its
// Payne-Hanek path for the upper bands and cost ~10 % of the kernel).
template <int L, int NF>
__device__ __forceinline__ void encode(const float (&v)[3], float (&feat)[NF]) {
    static_assert(NF >= 3 + 6 * L, "feature buffer too small");
    static_for<NF>([&](auto K) { feat[decltype(K)::value] = 0.0f; });
    feat[0] = v[0];
    feat[1] = v[1];
    feat[2] = v[2];
    static_for<3>([&](auto A) {
        constexpr int a = decltype(A)::value;
        double p = (double)v[a] * 0.15915494309189535;  // 1 / 2pi
        static_for<L>([&](auto B) {
            constexpr int b = decltype(B)::value;
            p = p - rint(p);                       // [-1/2, 1/2] turns, exact
            const double q = rint(p * 4.0);        // nearest quarter turn
            const float r = (float)(p - q * 0.25); // [-1/8, 1/8], exact before the conversion
            const int qi = (int)q & 3;
            float sn, cs;
            sincos_2pi_small(r, sn, cs);
            const float s_out = (qi & 1) ? cs : sn;
            const float c_out = (qi & 1) ? sn : cs;
            feat[3 + 6 * b + a] = (qi == 2 || qi == 3) ? -s_out : s_out;
            feat[3 + 6 * b + 3 + a] = (qi == 1 || qi == 2) ? -c_out : c_out;
            p = p + p;
        });
    });
}

enum { kModeRays = 0, kModeX = 1, kModePts = 2 };
struct MlpArgs {
    const float* wstream;
    const float* bias;
    const float* x;     // kModeX:    [n_points, 90] pre-embedded rows
    const float* rays;  // kModeRays: [n_rays, 11]
    const float* z;     // kModeRays: [n_rays, S]
    const float* pts;   // kModePts:  [n_points, 3]
    const float* dirs;  // kModePts:  [n_rays, 3] unit view directions
    long n_points;
    int S;
    float* raw;         // [n_points, 4]
    float* acts;        // training only: activation slab (act_off() matrices of p_pad rows), else null
    long p_pad;
};

constexpr int kMlpLds = kRingFrags * kFragBytes + kBiasFloats * 4;
constexpr int kStagePitch = 33;                       // 32x32 transpose tile, conflict-free
constexpr int kStageFloats = 32 * kStagePitch;
constexpr int kMlpLdsTrain = kMlpLds + 4 * kStageFloats * 4;

// Training: write NT post-ReLU accumulator tiles (channel = register, point = lane) as rows
// of a row-major [p_pad, ld] matrix: transpose each 32x32 tile through a per-wave LDS patch
// so that every global store instruction writes two full 128-byte row segments.
template <int NT>
__device__ __forceinline__ void save_tiles(const f32x16 (&t)[NT], float* dst, int ld, long p0, long n_points,
                                           float* stage, int lane) {
    const int m = lane & 31, h = lane >> 5;
    static_for<NT>([&](auto T) {
        constexpr int tt = decltype(T)::value;
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value;
            stage[m * kStagePitch + (r & 3) + 8 * (r >> 2) + 4 * h] = t[tt][r];
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<16>([&](auto RR) {
            constexpr int rr = decltype(RR)::value;
            const int row = 2 * rr + h;
            const float v = stage[row * kStagePitch + m];
            if (p0 + row < n_points) dst[(p0 + row) * ld + 32 * tt + m] = v;
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    });
}

template <int MODE, bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_f32_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_s = reinterpret_cast<float*>(smem + kRingFrags * kFragBytes);
    float* stage = reinterpret_cast<float*>(smem + kMlpLds) + (threadIdx.x >> 6) * kStageFloats;  // SAVE only

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;

    for (int i = tid; i < kBiasFloats; i += 256) bias_s[i] = a.bias[i];

    Diag dg;
    WStream ws;
    ws.dg = &dg;
    ws.gbase = reinterpret_cast<const char*>(a.wstream) + tid * 16;
    ws.gnext = ws.gbase;
    ws.next_slice = 0;
    ws.ring_wave = ring + wave * kFragBytes;
    ws.issue_rest<0, 0>();  // slice 0 -> slot 0

    FragReader fr;
    fr.addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    fr.addr1 = fr.addr0 + 64 * kFragBytes;
    const float* bias_h = bias_s + 4 * h;
    const long ntiles = (a.n_points + 127) >> 7;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        DIAG_ONLY(const unsigned long long t_tile = clock64(); dg.begin();)
        const long P = tile * 128 + wave * 32 + m;
        const bool valid = P < a.n_points;
        const long Pc = valid ? P : a.n_points - 1;

        // ---- inputs: this lane's half of the 64 point features and 32 direction features
        float pe[8][4], pd[4][4];
        if constexpr (MODE == kModeX) {
            const float* xr = a.x + Pc * (IDN_PTS_CH + IDN_VIEWS_CH);
            static_for<8>([&](auto G) {
                static_for<4>([&](auto J) {
                    constexpr int g = decltype(G)::value, j = decltype(J)::value;
                    const int k = 8 * g + 4 * h + j;
                    pe[g][j] = (k < IDN_PTS_CH) ? xr[k] : 0.0f;
                });
            });
            static_for<4>([&](auto G) {
                static_for<4>([&](auto J) {
                    constexpr int g = decltype(G)::value, j = decltype(J)::value;
                    const int k = 8 * g + 4 * h + j;
                    pd[g][j] = (k < IDN_VIEWS_CH) ? xr[IDN_PTS_CH + k] : 0.0f;
                });
            });
        } else {
            const long ray = Pc / a.S;
            float p[3], v[3];
            if constexpr (MODE == kModeRays) {
                const float* rr = a.rays + ray * IDN_RAY_FLOATS;
                const float zz = a.z[Pc];
                // pts = rays_o + rays_d * z, product and sum rounded separately
                // (audio_exp_nerf.py:332; this file is built with -ffp-contract=off)
                p[0] = rr[0] + rr[3] * zz;
                p[1] = rr[1] + rr[4] * zz;
                p[2] = rr[2] + rr[5] * zz;
                v[0] = rr[8];
                v[1] = rr[9];
                v[2] = rr[10];
            } else {
                p[0] = a.pts[Pc * 3 + 0];
                p[1] = a.pts[Pc * 3 + 1];
                p[2] = a.pts[Pc * 3 + 2];
                v[0] = a.dirs[ray * 3 + 0];
                v[1] = a.dirs[ray * 3 + 1];
                v[2] = a.dirs[ray * 3 + 2];
            }
            float fp[64], fd[32];
            encode<10, 64>(p, fp);
            encode<4, 32>(v, fd);
            static_for<8>([&](auto G) {
                static_for<4>([&](auto J) {
                    constexpr int g = decltype(G)::value, j = decltype(J)::value;
                    pe[g][j] = h ? fp[8 * g + 4 + j] : fp[8 * g + j];
                });
            });
            static_for<4>([&](auto G) {
                static_for<4>([&](auto J) {
                    constexpr int g = decltype(G)::value, j = decltype(J)::value;
                    pd[g][j] = h ? fd[8 * g + 4 + j] : fd[8 * g + j];
                });
            });
        }
        const long p0 = tile * 128 + wave * 32;  // first point of this wave
        if constexpr (SAVE) {
            if (valid) {
                float* x0 = a.acts + act_off(kActX0) * a.p_pad + P * 64 + 4 * h;
                float* dr = a.acts + act_off(kActDir) * a.p_pad + P * 64 + 4 * h;
                static_for<8>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    f32x4 v = {pe[g][0], pe[g][1], pe[g][2], pe[g][3]};
                    *reinterpret_cast<f32x4*>(x0 + 8 * g) = v;
                });
                static_for<8>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (g < 4) v = f32x4{pd[g][0], pd[g][1], pd[g][2], pd[g][3]};
                    *reinterpret_cast<f32x4*>(dr + 8 * g) = v;
                });
            }
        }
        DIAG_END(dg, kDgInput);

        // Two sets of eight 32x32 tiles take turns as a layer's input (B operands) and output
        // (accumulators); a finished layer's output is ReLU'd in place and read by the next.
        f32x16 A[8], B[8], V[5];
        auto pe_get = [&](auto G, auto J) { return pe[decltype(G)::value][decltype(J)::value]; };
        auto tiles_get = [](f32x16* arr) {
            return [arr](auto G, auto J) {
                constexpr int g = decltype(G)::value, j = decltype(J)::value;
                return arr[g >> 2][(g & 3) * 4 + j];
            };
        };
        // One layer of the trunk / colour branch.  Inference: bias of tile 0 up front, everything
        // else in MFMA shadows (LayerSide); the last tile's ReLU is owed to the next layer.
        // Training (SAVE): plain boundaries -- bias, MFMAs, ReLU, store the activations.
        auto layer = [&](auto F0c, auto NTc, auto KGc, auto DEFERc, auto& out, f32x16* deferred, auto&& bget,
                         const float* bias_l, int save_idx) {
            constexpr int F0 = decltype(F0c)::value, NT = decltype(NTc)::value, KG = decltype(KGc)::value;
            constexpr bool DEFER = decltype(DEFERc)::value != 0;
            DIAG_BEGIN(dg);
            if constexpr (SAVE) load_bias<NT>(out, bias_l);
            else bias_tile(out[0], bias_l);
            DIAG_END(dg, kDgBoundary);
            if constexpr (SAVE) {
                run_layer<F0, NT, KG>(out, bget, ws, fr, NoSide{});
                DIAG_BEGIN(dg);
                if (save_idx >= 0) {  // hidden layer: ReLU + record
                    relu_tiles<NT>(out);
                    save_tiles<NT>(out, a.acts + (long)act_off(save_idx) * a.p_pad, 32 * NT, p0, a.n_points, stage, lane);
                }
                DIAG_END(dg, kDgBoundary);
            } else {
                run_layer<F0, NT, KG>(out, bget, ws, fr, LayerSide<NT, KG / 2, DEFER>{&out[0], deferred, bias_l});
            }
        };
        constexpr bool D = !SAVE;  // deferred last-tile ReLU only exists on the inference path

        // ---- pts_linears.0 : PE(64) -> 256
        layer(ic<layer_f0(0)>{}, ic<8>{}, ic<8>{}, ic<0>{}, A, nullptr, pe_get, bias_h + bias_off(0), kActA1 + 0);
        // ---- pts_linears.1..4 : 256 -> 256   (A -> B -> A -> B -> A)
#pragma unroll 1
        for (int l = 1; l <= 3; l += 2) {
            layer(ic<layer_f0(1)>{}, ic<8>{}, ic<32>{}, ic<D>{}, B, &A[7], tiles_get(A), bias_h + l * 256, kActA1 + l);
            layer(ic<layer_f0(2)>{}, ic<8>{}, ic<32>{}, ic<D>{}, A, &B[7], tiles_get(B), bias_h + (l + 1) * 256, kActA1 + l + 1);
        }
        // ---- pts_linears.5 : [PE(64) | 256] -> 256   (skip connection, face_nerf.py:61-62)
        layer(ic<layer_f0(5)>{}, ic<8>{}, ic<40>{}, ic<D>{}, B, &A[7],
              [&](auto G, auto J) {
                  constexpr int g = decltype(G)::value, j = decltype(J)::value;
                  if constexpr (g < 8) return pe[g][j];
                  else return A[(g - 8) >> 2][((g - 8) & 3) * 4 + j];
              },
              bias_h + bias_off(5), kActA1 + 5);
        // ---- pts_linears.6, .7
        layer(ic<layer_f0(6)>{}, ic<8>{}, ic<32>{}, ic<D>{}, A, &B[7], tiles_get(B), bias_h + bias_off(6), kActA1 + 6);
        layer(ic<layer_f0(7)>{}, ic<8>{}, ic<32>{}, ic<D>{}, B, &A[7], tiles_get(A), bias_h + bias_off(7), kActA1 + 7);
        // ---- views_linears.0 (+ alpha_linear as channel 128) : [256 | dirPE(32)] -> 160
        //      tiles 0..3 are hidden units (ReLU'd while the next tile accumulates), tile 4 row 0 is sigma
        layer(ic<layer_f0(8)>{}, ic<5>{}, ic<36>{}, ic<D>{}, V, &B[7],
              [&](auto G, auto J) {
                  constexpr int g = decltype(G)::value, j = decltype(J)::value;
                  if constexpr (g < 32) return B[g >> 2][(g & 3) * 4 + j];
                  else return pd[g - 32][j];
              },
              bias_h + bias_off(8), -1);
        const float sigma = V[4][0];  // channel 128 = tile 4, register 0, lane half 0
        if constexpr (SAVE) {
            DIAG_BEGIN(dg);
            f32x16(&V4)[4] = reinterpret_cast<f32x16(&)[4]>(V);
            relu_tiles<4>(V4);
            save_tiles<4>(V4, a.acts + (long)act_off(kActV1) * a.p_pad, 128, p0, a.n_points, stage, lane);
            DIAG_END(dg, kDgBoundary);
        }
        // ---- views_linears.1, .2 : 128 -> 128   (V -> A[0..3] -> V[0..3])
        f32x16(&A4)[4] = reinterpret_cast<f32x16(&)[4]>(A);
        f32x16(&V4b)[4] = reinterpret_cast<f32x16(&)[4]>(V);
        layer(ic<layer_f0(9)>{}, ic<4>{}, ic<16>{}, ic<0>{}, A4, nullptr, tiles_get(V), bias_h + bias_off(9), kActV1 + 1);
        layer(ic<layer_f0(10)>{}, ic<4>{}, ic<16>{}, ic<D>{}, V4b, &A[3], tiles_get(A), bias_h + bias_off(10), kActV1 + 2);
        // ---- rgb_linear : 128 -> 3 (rows 0..2 of one tile)
        f32x16 rgb[1];
        layer(ic<layer_f0(11)>{}, ic<1>{}, ic<16>{}, ic<D>{}, rgb, &V[3], tiles_get(V), bias_h + bias_off(11), -1);
        finish_pass<kUsedFrags>(ws);

        DIAG_BEGIN(dg);
        if (valid && h == 0) {
            f32x4 o;
            o.x = rgb[0][0];
            o.y = rgb[0][1];
            o.z = rgb[0][2];
            o.w = sigma;
            *reinterpret_cast<f32x4*>(a.raw + P * 4) = o;
        }
        DIAG_END(dg, kDgStore);
        DIAG_ONLY(dg.acc[kDgTotal] += clock64() - t_tile;)
    }
#ifdef IDN_DIAG
    if (lane == 0)
        for (int c = 0; c < 5; ++c) atomicAdd(&g_diag[c], dg.acc[c]);
    if (lane == 0) atomicAdd(&g_diag[5], 1ull);
#endif
    // drain the slice prefetched for a pass that will not happen
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}

int launch_mlp_f32(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                   const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s,
                   float* acts, int64_t p_pad) {
    if (n_points <= 0) return IDN_OK;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        IDN_HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        IDN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount;
        IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModeRays, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
        IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModeX, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
        IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModePts, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
        IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModeRays, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLdsTrain));
    }
    const int64_t ntiles = (n_points + 127) / 128;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    MlpArgs a{packed, folded, x, rays, z, pts, dirs, (long)n_points, n_samples, raw, acts, (long)p_pad};
    ProfScope prof(s, n_points);
    if (acts) {
        if (x || pts) return fail(IDN_EUNSUPPORTED, "activation saving is only built for the rays+z input mode");
        hipLaunchKernelGGL((mlp_f32_kernel<kModeRays, true>), dim3(grid), dim3(256), kMlpLdsTrain, s, a);
    } else if (x)
        hipLaunchKernelGGL((mlp_f32_kernel<kModeX, false>), dim3(grid), dim3(256), kMlpLds, s, a);
    else if (pts)
        hipLaunchKernelGGL((mlp_f32_kernel<kModePts, false>), dim3(grid), dim3(256), kMlpLds, s, a);
    else
        hipLaunchKernelGGL((mlp_f32_kernel<kModeRays, false>), dim3(grid), dim3(256), kMlpLds, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

#ifdef IDN_DIAG
extern "C" int idealnerf_diag_read(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_diag), 8 * sizeof(unsigned long long)) != hipSuccess) return -3;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof(z)) != hipSuccess) return -3;
    return 0;
}
#endif

}  // namespace idn
