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
// ---------------------------------------------------------------------------
struct WStream {
    Diag* dg;
    const char* gbase;  // stream start + this lane's 16-byte column
    const char* gnext;  // same, for the next slice to fetch
    int next_slice;
    char* ring_wave;    // ring + wave * 1 KiB (wave-uniform LDS destination base)

    // Fetch the next 64 KiB slice into ring slot SLOT: 16 x (4 waves x 1 KiB).
    template <int SLOT>
    __device__ __forceinline__ void issue() {
        static_for<kSliceFrags / 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gnext + i * 4096),
                                             LDS_PTR(ring_wave + SLOT * kSliceBytes + i * 4096), 16, 0, 0);
        });
        gnext += kSliceBytes;
        if (++next_slice == kNumSlices) {
            next_slice = 0;
            gnext = gbase;
        }
    }
    // Enter the slice that lives in slot SLOT: every wave's share of it has landed
    // (vmcnt(0) precedes the barrier) and every wave is done reading the other slot.
    template <int SLOT>
    __device__ __forceinline__ void enter() {
        __syncthreads();
        issue<SLOT ^ 1>();
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_other() {
        issue<SLOT ^ 1>();
    }
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// A-fragment reads.  hipcc (ROCm 7.2) waits lgkmcnt(0) after a prefetching ds_read --
// i.e. for the read it has just issued -- which exposes one LDS latency per eight MFMAs
// (measured: SQ_WAIT_ANY 10 % of wave cycles, MFMA pipe 87 % busy).  The reads are therefore
// issued from inline asm, which the compiler does not count, and retired by a counted wait
// tied to the destination registers ("+v"): LDS operations return in order, so
// lgkmcnt(1) right after issuing fragment i+1 means fragment i has landed.  Compiler-issued
// LDS/SMEM operations in between only make these waits stricter (never weaker), and its own
// counted waits likewise (cdna_hip_programming.md section 5.7).
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

template <int F>
__device__ __forceinline__ void enter_slice_of(WStream& ws) {
    if constexpr (((F / kSliceFrags) & 1) == 0) ws.template enter<0>();
    else ws.template enter<1>();
}
// after the barrier that opens the slice of fragment F and after the first fragment reads:
// start fetching the following slice into the other slot
template <int F>
__device__ __forceinline__ void prefetch_after(WStream& ws) {
    if constexpr (((F / kSliceFrags) & 1) == 0) ws.template prefetch_other<0>();
    else ws.template prefetch_other<1>();
}

// acc[t] += W(layer)[32t.., k-group g] . B(g)   for all g, t of one layer.
// F0 = index of the layer's first fragment in the stream (only F0 mod ring matters for
// addressing, so layers whose F0 agree mod kRingFrags can share one instantiation).
// On entry fr.pref holds fragment F0 in flight unless F0 opens a slice; on exit it holds
// fragment F0 + NT*KG in flight unless that one opens a slice.
template <int F0, int NT, int KG, class BGet>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[NT], BGet&& bget, WStream& ws, FragReader& fr) {
    // Fragments are consumed in pairs (2p, 2p+1): for NT >= 2 they belong to different
    // accumulator tiles, so the eight MFMAs of a pair alternate between two independent
    // accumulation chains (a dependent v_mfma_f32_32x32x2_f32 issued back to back costs a
    // few cycles more than its 64-cycle issue interval).  The reads of pair p+1 are issued
    // ahead of the MFMAs of pair p.  Slice boundaries (multiples of 64) never split a pair.
    constexpr int N = NT * KG;
    static_assert(N % 2 == 0 && F0 % 2 == 0, "fragments are consumed in pairs");
    if constexpr (F0 % kSliceFrags == 0) {
        DIAG_BEGIN(*ws.dg);
        __syncthreads();  // slice published; the other slot is free
        DIAG_END(*ws.dg, kDgBarrier);
        fr.pref0 = fr.template issue<F0>();
        fr.pref1 = fr.template issue<F0 + 1>();
        prefetch_after<F0>(ws);  // address generation + 16 glds ride under the LDS latency
    }
    f32x4 a0 = fr.pref0, a1 = fr.pref1;
    static_for<N / 2>([&](auto PI) {
        constexpr int i0 = 2 * decltype(PI)::value, i1 = i0 + 1;
        constexpr int g0 = i0 / NT, t0 = i0 % NT, g1 = i1 / NT, t1 = i1 % NT;
        constexpr int f = F0 + i0;
        constexpr bool next_crosses = ((f + 2) % kSliceFrags == 0);
        f32x4 n0 = a0, n1 = a1;
        if constexpr (!next_crosses) {
            n0 = fr.template issue<f + 2>();
            n1 = fr.template issue<f + 3>();
            FragReader::retire<2>(a0, a1);
        } else {
            FragReader::retire<0>(a0, a1);
        }
        acc[t0] = mfma(a0.x, bget(ic<g0>{}, ic<0>{}), acc[t0]);
        acc[t1] = mfma(a1.x, bget(ic<g1>{}, ic<0>{}), acc[t1]);
        acc[t0] = mfma(a0.y, bget(ic<g0>{}, ic<1>{}), acc[t0]);
        acc[t1] = mfma(a1.y, bget(ic<g1>{}, ic<1>{}), acc[t1]);
        acc[t0] = mfma(a0.z, bget(ic<g0>{}, ic<2>{}), acc[t0]);
        acc[t1] = mfma(a1.z, bget(ic<g1>{}, ic<2>{}), acc[t1]);
        acc[t0] = mfma(a0.w, bget(ic<g0>{}, ic<3>{}), acc[t0]);
        acc[t1] = mfma(a1.w, bget(ic<g1>{}, ic<3>{}), acc[t1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (next_crosses && i0 + 2 < N) {
            DIAG_BEGIN(*ws.dg);
            __syncthreads();
            DIAG_END(*ws.dg, kDgBarrier);
            n0 = fr.template issue<f + 2>();
            n1 = fr.template issue<f + 3>();
            prefetch_after<f + 2>(ws);
        }
        a0 = n0;
        a1 = n1;
    });
    fr.pref0 = a0;  // pair F0+N (already in flight) when it does not open a slice
    fr.pref1 = a1;
}

// Skip the stream forward over slices that hold only padding (end of a pass).
template <int F_END>
__device__ __forceinline__ void finish_pass(WStream& ws) {
    constexpr int consumed = (F_END + kSliceFrags - 1) / kSliceFrags;
    static_for<kNumSlices - consumed>([&](auto I) {
        constexpr int s = consumed + decltype(I)::value;
        if constexpr ((s & 1) == 0) ws.template enter<0>();
        else ws.template enter<1>();
    });
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NT], const float* bias_half /* bias_s + off + 4h */) {
    static_for<NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        static_for<4>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias_half + 32 * t + 8 * q);
            acc[t][4 * q + 0] = b.x;
            acc[t][4 * q + 1] = b.y;
            acc[t][4 * q + 2] = b.z;
            acc[t][4 * q + 3] = b.w;
        });
    });
}

template <int NT>
__device__ __forceinline__ void relu_to(f32x16 (&dst)[NT], const f32x16 (&src)[NT]) {
    static_for<NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value;
            dst[t][r] = __builtin_amdgcn_fmed3f(src[t][r], 0.0f, __builtin_inff());  // one v_med3_f32
        });
    });
}

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
    ws.issue<0>();

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
        auto save_hid = [&](const f32x16 (&t)[8], int layer /*1..8*/) {
            if constexpr (SAVE) save_tiles<8>(t, a.acts + act_off(kActA1 + layer - 1) * a.p_pad, 256, p0, a.n_points, stage, lane);
        };
        auto save_hv = [&](const f32x16 (&t)[4], int layer /*1..3*/) {
            if constexpr (SAVE) save_tiles<4>(t, a.acts + act_off(kActV1 + layer - 1) * a.p_pad, 128, p0, a.n_points, stage, lane);
        };

        auto pe_get = [&](auto G, auto J) { return pe[decltype(G)::value][decltype(J)::value]; };

        f32x16 acc[8], hid[8];
        auto hid_get = [&](auto G, auto J) {
            constexpr int g = decltype(G)::value, j = decltype(J)::value;
            return hid[g >> 2][(g & 3) * 4 + j];
        };

        DIAG_END(dg, kDgInput);
        // ---- pts_linears.0 : PE(64) -> 256
        DIAG_BEGIN(dg);
        load_bias<8>(acc, bias_h + bias_off(0));
        DIAG_END(dg, kDgBoundary);
        run_layer<layer_f0(0), 8, 8>(acc, pe_get, ws, fr);
        DIAG_BEGIN(dg);
        relu_to<8>(hid, acc);
        DIAG_END(dg, kDgBoundary);
        DIAG_BEGIN(dg);
        save_hid(hid, 1);
        DIAG_END(dg, kDgBoundary);
        // ---- pts_linears.1..4 : 256 -> 256
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
            DIAG_BEGIN(dg);
            load_bias<8>(acc, bias_h + l * 256);
            DIAG_END(dg, kDgBoundary);
            run_layer<layer_f0(1), 8, 32>(acc, hid_get, ws, fr);
            DIAG_BEGIN(dg);
            relu_to<8>(hid, acc);
            DIAG_END(dg, kDgBoundary);
            DIAG_BEGIN(dg);
            save_hid(hid, l + 1);
            DIAG_END(dg, kDgBoundary);
        }
        // ---- pts_linears.5 : [PE(64) | 256] -> 256   (skip connection, face_nerf.py:61-62)
        DIAG_BEGIN(dg);
        load_bias<8>(acc, bias_h + bias_off(5));
        DIAG_END(dg, kDgBoundary);
        run_layer<layer_f0(5), 8, 40>(
            acc,
            [&](auto G, auto J) {
                constexpr int g = decltype(G)::value, j = decltype(J)::value;
                if constexpr (g < 8) return pe[g][j];
                else return hid[(g - 8) >> 2][((g - 8) & 3) * 4 + j];
            },
            ws, fr);
        DIAG_BEGIN(dg);
        relu_to<8>(hid, acc);
        DIAG_END(dg, kDgBoundary);
        DIAG_BEGIN(dg);
        save_hid(hid, 6);
        DIAG_END(dg, kDgBoundary);
        // ---- pts_linears.6..7
#pragma unroll 1
        for (int l = 6; l <= 7; ++l) {
            DIAG_BEGIN(dg);
            load_bias<8>(acc, bias_h + l * 256);
            DIAG_END(dg, kDgBoundary);
            run_layer<layer_f0(6), 8, 32>(acc, hid_get, ws, fr);
            DIAG_BEGIN(dg);
            relu_to<8>(hid, acc);
            DIAG_END(dg, kDgBoundary);
            DIAG_BEGIN(dg);
            save_hid(hid, l + 1);
            DIAG_END(dg, kDgBoundary);
        }
        // ---- views_linears.0 (+ alpha_linear as channel 128) : [256 | dirPE(32)] -> 160
        f32x16 va[5];
        DIAG_BEGIN(dg);
        load_bias<5>(va, bias_h + bias_off(8));
        DIAG_END(dg, kDgBoundary);
        run_layer<layer_f0(8), 5, 36>(
            va,
            [&](auto G, auto J) {
                constexpr int g = decltype(G)::value, j = decltype(J)::value;
                if constexpr (g < 32) return hid[g >> 2][(g & 3) * 4 + j];
                else return pd[g - 32][j];
            },
            ws, fr);
        const float sigma = va[4][0];  // channel 128 = tile 4, register 0, lane half 0
        f32x16 hv[4], vb[4];
        static_for<4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            static_for<16>([&](auto R) { hv[t][decltype(R)::value] = __builtin_amdgcn_fmed3f(va[t][decltype(R)::value], 0.0f, __builtin_inff()); });
        });
        DIAG_BEGIN(dg);
        save_hv(hv, 1);
        DIAG_END(dg, kDgBoundary);
        auto hv_get = [&](auto G, auto J) {
            constexpr int g = decltype(G)::value, j = decltype(J)::value;
            return hv[g >> 2][(g & 3) * 4 + j];
        };
        // ---- views_linears.1, .2 : 128 -> 128
        DIAG_BEGIN(dg);
        load_bias<4>(vb, bias_h + bias_off(9));
        DIAG_END(dg, kDgBoundary);
        run_layer<layer_f0(9), 4, 16>(vb, hv_get, ws, fr);
        DIAG_BEGIN(dg);
        relu_to<4>(hv, vb);
        DIAG_END(dg, kDgBoundary);
        DIAG_BEGIN(dg);
        save_hv(hv, 2);
        DIAG_END(dg, kDgBoundary);
        DIAG_BEGIN(dg);
        load_bias<4>(vb, bias_h + bias_off(10));
        DIAG_END(dg, kDgBoundary);
        run_layer<layer_f0(10), 4, 16>(vb, hv_get, ws, fr);
        DIAG_BEGIN(dg);
        relu_to<4>(hv, vb);
        DIAG_END(dg, kDgBoundary);
        DIAG_BEGIN(dg);
        save_hv(hv, 3);
        DIAG_END(dg, kDgBoundary);
        // ---- rgb_linear : 128 -> 3 (rows 0..2 of one tile)
        f32x16 rgb[1];
        DIAG_BEGIN(dg);
        load_bias<1>(rgb, bias_h + bias_off(11));
        DIAG_END(dg, kDgBoundary);
        run_layer<layer_f0(11), 1, 16>(rgb, hv_get, ws, fr);
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
