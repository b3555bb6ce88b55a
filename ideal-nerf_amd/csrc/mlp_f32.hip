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

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------
// weight stream: global -> LDS ring
// ---------------------------------------------------------------------------
struct WStream {
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
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc[t] += W(layer)[32t.., k-group g] . B(g)   for all g, t of one layer.
// F0 = index of the layer's first fragment in the stream (only F0 mod ring matters for
// addressing, so layers whose F0 agree mod kRingFrags can share one instantiation).
template <int F>
__device__ __forceinline__ void enter_slice_of(WStream& ws) {
    if constexpr (((F / kSliceFrags) & 1) == 0) ws.template enter<0>();
    else ws.template enter<1>();
}
template <int F>
__device__ __forceinline__ f32x4 read_frag(const char* ring_lane) {
    return *reinterpret_cast<const f32x4*>(ring_lane + (F % kRingFrags) * kFragBytes);
}

template <int F0, int NT, int KG, class BGet>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[NT], BGet&& bget, WStream& ws, const char* ring_lane) {
    // One fragment (ds_read_b128) feeds four MFMAs (256 cycles); the read for fragment
    // i+1 is issued ahead of the MFMAs of fragment i so its LDS latency is covered.  At
    // a slice boundary the read has to wait for the barrier that publishes the slice.
    constexpr int N = NT * KG;
    if constexpr (F0 % kSliceFrags == 0) enter_slice_of<F0>(ws);
    f32x4 a_cur = read_frag<F0>(ring_lane);
    static_for<N>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int g = i / NT, t = i % NT;
        constexpr int f = F0 + i;
        constexpr bool has_next = (i + 1 < N);
        constexpr bool next_crosses = ((f + 1) % kSliceFrags == 0);
        f32x4 a_next = a_cur;
        if constexpr (has_next && !next_crosses) {
            a_next = read_frag<f + 1>(ring_lane);
            __builtin_amdgcn_sched_barrier(0);  // keep the read ahead of this fragment's MFMAs
        }
        acc[t] = mfma(a_cur.x, bget(ic<g>{}, ic<0>{}), acc[t]);
        acc[t] = mfma(a_cur.y, bget(ic<g>{}, ic<1>{}), acc[t]);
        acc[t] = mfma(a_cur.z, bget(ic<g>{}, ic<2>{}), acc[t]);
        acc[t] = mfma(a_cur.w, bget(ic<g>{}, ic<3>{}), acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (has_next && next_crosses) {
            enter_slice_of<f + 1>(ws);
            a_next = read_frag<f + 1>(ring_lane);
        }
        a_cur = a_next;
    });
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
            dst[t][r] = fmaxf(src[t][r], 0.0f);
        });
    });
}

// gamma_L(v) for a 3-vector, as 3 + 6L features in the reference's order
// (helper.py:183-201): [v, sin(2^0 v), cos(2^0 v), ..., sin(2^(L-1) v), cos(2^(L-1) v)],
// zero padded to NF.  2^b * v is exact, sincosf is the accurate (range-reducing) one.
template <int L, int NF>
__device__ __forceinline__ void encode(const float (&v)[3], float (&feat)[NF]) {
    static_assert(NF >= 3 + 6 * L, "feature buffer too small");
    static_for<NF>([&](auto K) { feat[decltype(K)::value] = 0.0f; });
    feat[0] = v[0];
    feat[1] = v[1];
    feat[2] = v[2];
    static_for<L>([&](auto B) {
        constexpr int b = decltype(B)::value;
        constexpr float freq = (float)(1 << b);
        static_for<3>([&](auto A) {
            constexpr int a = decltype(A)::value;
            float s, c;
            sincosf(v[a] * freq, &s, &c);
            feat[3 + 6 * b + a] = s;
            feat[3 + 6 * b + 3 + a] = c;
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

    WStream ws;
    ws.gbase = reinterpret_cast<const char*>(a.wstream) + tid * 16;
    ws.gnext = ws.gbase;
    ws.next_slice = 0;
    ws.ring_wave = ring + wave * kFragBytes;
    ws.issue<0>();

    const char* ring_lane = ring + lane * 16;
    const float* bias_h = bias_s + 4 * h;
    const long ntiles = (a.n_points + 127) >> 7;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
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

        // ---- pts_linears.0 : PE(64) -> 256
        load_bias<8>(acc, bias_h + bias_off(0));
        run_layer<layer_f0(0), 8, 8>(acc, pe_get, ws, ring_lane);
        relu_to<8>(hid, acc);
        save_hid(hid, 1);
        // ---- pts_linears.1..4 : 256 -> 256
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
            load_bias<8>(acc, bias_h + l * 256);
            run_layer<layer_f0(1), 8, 32>(acc, hid_get, ws, ring_lane);
            relu_to<8>(hid, acc);
            save_hid(hid, l + 1);
        }
        // ---- pts_linears.5 : [PE(64) | 256] -> 256   (skip connection, face_nerf.py:61-62)
        load_bias<8>(acc, bias_h + bias_off(5));
        run_layer<layer_f0(5), 8, 40>(
            acc,
            [&](auto G, auto J) {
                constexpr int g = decltype(G)::value, j = decltype(J)::value;
                if constexpr (g < 8) return pe[g][j];
                else return hid[(g - 8) >> 2][((g - 8) & 3) * 4 + j];
            },
            ws, ring_lane);
        relu_to<8>(hid, acc);
        save_hid(hid, 6);
        // ---- pts_linears.6..7
#pragma unroll 1
        for (int l = 6; l <= 7; ++l) {
            load_bias<8>(acc, bias_h + l * 256);
            run_layer<layer_f0(6), 8, 32>(acc, hid_get, ws, ring_lane);
            relu_to<8>(hid, acc);
            save_hid(hid, l + 1);
        }
        // ---- views_linears.0 (+ alpha_linear as channel 128) : [256 | dirPE(32)] -> 160
        f32x16 va[5];
        load_bias<5>(va, bias_h + bias_off(8));
        run_layer<layer_f0(8), 5, 36>(
            va,
            [&](auto G, auto J) {
                constexpr int g = decltype(G)::value, j = decltype(J)::value;
                if constexpr (g < 32) return hid[g >> 2][(g & 3) * 4 + j];
                else return pd[g - 32][j];
            },
            ws, ring_lane);
        const float sigma = va[4][0];  // channel 128 = tile 4, register 0, lane half 0
        f32x16 hv[4], vb[4];
        static_for<4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            static_for<16>([&](auto R) { hv[t][decltype(R)::value] = fmaxf(va[t][decltype(R)::value], 0.0f); });
        });
        save_hv(hv, 1);
        auto hv_get = [&](auto G, auto J) {
            constexpr int g = decltype(G)::value, j = decltype(J)::value;
            return hv[g >> 2][(g & 3) * 4 + j];
        };
        // ---- views_linears.1, .2 : 128 -> 128
        load_bias<4>(vb, bias_h + bias_off(9));
        run_layer<layer_f0(9), 4, 16>(vb, hv_get, ws, ring_lane);
        relu_to<4>(hv, vb);
        save_hv(hv, 2);
        load_bias<4>(vb, bias_h + bias_off(10));
        run_layer<layer_f0(10), 4, 16>(vb, hv_get, ws, ring_lane);
        relu_to<4>(hv, vb);
        save_hv(hv, 3);
        // ---- rgb_linear : 128 -> 3 (rows 0..2 of one tile)
        f32x16 rgb[1];
        load_bias<1>(rgb, bias_h + bias_off(11));
        run_layer<layer_f0(11), 1, 16>(rgb, hv_get, ws, ring_lane);
        finish_pass<kUsedFrags>(ws);

        if (valid && h == 0) {
            f32x4 o;
            o.x = rgb[0][0];
            o.y = rgb[0][1];
            o.z = rgb[0][2];
            o.w = sigma;
            *reinterpret_cast<f32x4*>(a.raw + P * 4) = o;
        }
    }
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

}  // namespace idn
