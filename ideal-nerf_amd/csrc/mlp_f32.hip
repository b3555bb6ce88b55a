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
#include "mlp_f32_layers.h"

namespace idn {

// Training: the NT post-ReLU accumulator tiles of a layer (channel = register, point = lane) are
// written as rows of a row-major [p_pad, ld] matrix, in the shadow of the MFMAs: tile t-1 is ReLU'd
// at pair-step 0 of tile t, scattered into a per-wave 32x33 LDS patch (step 1) and written out as
// two full 128-byte row segments per store, the four values of quad q being READ from the patch at
// step 2 + q and STORED at step 3 + q.  The patch reads are inline asm like the fragment reads: LDS
// returns in order, so the pair-step's counted wait has covered them by the time they are stored
// (as plain loads each one was followed by `lgkmcnt(0)`: sixteen exposed round trips per tile).
// Every row of the p_pad-row slab is written -- padding rows repeat the last point -- because the
// weight-gradient GEMMs contract over all p_pad rows (their deltas are zero, but 0 x garbage is not).
template <int NT, int STEPS, int LD>
struct SaveSide {
    static constexpr bool kShadowStore = STEPS >= 8;
    uint32_t* mk;     // [4] ReLU mask bits of this layer, collected tile by tile
    f32x16* out;
    __amdgpu_buffer_rsrc_t rsrc;   // this wave's 32 rows of the activation matrix (LD floats per row)
    uint32_t voff;                 // byte offset of [row h][column m]
    float* stage;     // this wave's transpose patch
    float* rb;        // [4] row values in flight between their read and their store
    uint32_t raddr;   // LDS byte address of patch[h][m]
    int m, h;
    template <int T>
    __device__ __forceinline__ void scatter(ic<T>) const {
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value;
            stage[m * kStagePitch + (r & 3) + 8 * (r >> 2) + 4 * h] = out[T][r];
        });
        asm volatile("" ::: "memory");   // the asm reads below come after these writes (a wave's LDS operations execute in order)
    }
    template <int Q>
    __device__ __forceinline__ void rows_read(ic<Q>) const {   // rows 2 (4Q + i) + h, i = 0..3
        static_for<4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rb[i]) : "v"(raddr), "n"(2 * (4 * Q + i) * kStagePitch * 4) : "memory");
        });
    }
    template <int T, int Q>
    __device__ __forceinline__ void rows_store(ic<T>, ic<Q>) const {
        asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));   // not before the wait that precedes this call
#ifdef IDN_TIMING_NO_ROW_STORES   // timing-only experiment (wrong results): what do the row stores cost?
        return;
#endif
        static_for<4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            // descriptor + one lane-offset VGPR + a compile-time scalar offset: no per-row address arithmetic or registers
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rb[i]), rsrc, voff, (2 * (4 * Q + i) * LD + 32 * T) * 4, 0);
        });
    }
    template <int T>
    __device__ __forceinline__ void flush_tile(ic<T>) const {   // outside the pair-step pipeline: explicit waits
#ifdef IDN_TIMING_NO_FLUSH   // timing-only experiment (wrong results): what do the exposed layer-end flushes cost?
        return;
#endif
        scatter(ic<T>{});
        static_for<4>([&](auto Q) {
            rows_read(Q);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3])::"memory");
            rows_store(ic<T>{}, Q);
        });
    }
    __device__ __forceinline__ void finish() const {   // after the layer: what the shadow schedule did not cover
        if constexpr (!kShadowStore) static_for<NT - 1>([&](auto T) { collect_signs<decltype(T)::value>(out[decltype(T)::value], mk); });
        collect_signs<NT - 1>(out[NT - 1], mk);   // tiles in order: the bits of a dword are pushed oldest first
        relu_regs<0, 16>(out[NT - 1]);
        if constexpr (kShadowStore) flush_tile(ic<NT - 1>{});
        else {
            static_for<NT - 1>([&](auto T) { relu_regs<0, 16>(out[decltype(T)::value]); });
            static_for<NT>([&](auto T) { flush_tile(T); });
        }
    }
    template <int T, int S, int H>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>, ic<H>) const {
        if constexpr (kShadowStore && T > 0 && H == 0) {
            if constexpr (S == 0) {
                collect_signs<T - 1>(out[T - 1], mk);
                relu_regs<0, 16>(out[T - 1]);
            }
            if constexpr (S == 1) scatter(ic<T - 1>{});
            if constexpr (S >= 3 && S <= 6) rows_store(ic<T - 1>{}, ic<S - 3>{});
            if constexpr (S >= 2 && S <= 5) rows_read(ic<S - 2>{});
        }
    }
};

template <int MODE, bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_f32_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_s = reinterpret_cast<float*>(smem + kRingFrags * kFragBytes);
    float* stage = reinterpret_cast<float*>(smem + kMlpLds) + (threadIdx.x >> 6) * kStageFloats;  // SAVE only
    const uint32_t raddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)stage +
                           (((threadIdx.x & 63) >> 5) * kStagePitch + (threadIdx.x & 31)) * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;

    for (int i = tid; i < kBiasFloats; i += 256) bias_s[i] = a.bias[i];
    __syncthreads();  // the bias block is read (by other waves) before the first slice barrier

    Diag dg;
    WStream ws;
    ws.dg = &dg;
    ws.init(a.wstream, kNumSlices, ring, tid, wave);
    PeLane pln;
    pln.init(h);

    FragReader fr;
    fr.addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    fr.addr1 = fr.addr0 + 64 * kFragBytes;
    const float* bias_h = bias_s + 4 * h;
    const long ntiles = (a.n_points + 127) >> 7;

    PointIn cur, nxt;   // raw inputs of this lane's point, loaded one pass ahead (mlp_common.h)
    load_point<MODE>(a, blockIdx.x, wave, m, cur);
    nxt = cur;
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
            float p[3], v[3];
            point_of<MODE>(cur, p, v);
            PeAxes axp, axd;
            axp.init(p, h);
            axd.init(v, h);
            static_for<8>([&](auto G) {
                static_for<4>([&](auto J) {
                    constexpr int g = decltype(G)::value, j = decltype(J)::value;
                    pe[g][j] = pe_slot<8 * g + j, 10>(axp, pln);
                    if constexpr (g < 4) pd[g][j] = pe_slot<8 * g + j, 4>(axd, pln);
                });
            });
        }
        const long p0 = tile * 128 + wave * 32;  // first point of this wave
        if constexpr (SAVE) {
            {   // every row of the slab, padding included (see SaveSide)
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

        float rgb[3], sigma;
        if constexpr (!SAVE) {
            // the next tile's point inputs: loaded after pts_linears.5's first slice opens, touched one layer later
            f32_inference_pass(pe, pd, bias_s, bias_h, ws, fr,
                               [&]() { if constexpr (MODE != kModeX) load_point<MODE>(a, tile + gridDim.x, wave, m, nxt); },
                               [&]() { if constexpr (MODE != kModeX) touch_point(nxt); }, rgb, sigma);
        } else {
            // Training on the fp32 pipe (IDN_TRAIN_PRECISION=f32): the same layers with plain boundaries -- bias, MFMAs, ReLU -- and
            // every hidden layer's activations and ReLU mask recorded in the MFMA shadow (SaveSide).
            f32x16 A[8], B[8], V[5];
            float rb[4];
            uint32_t mk[4] = {0u, 0u, 0u, 0u};   // the current layer's ReLU mask bits
            auto store_mask = [&](int id) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4* mp = reinterpret_cast<u32x4*>(a.acts + (size_t)kActCols * a.p_pad) + mask_index(id, a.p_pad, tile * 4 + wave, lane);
                *mp = u32x4{mk[0], mk[1], mk[2], mk[3]};
            };
            auto pe_get = [&](auto G, auto J) { return pe[decltype(G)::value][decltype(J)::value]; };
            auto tiles_get = [](f32x16* arr) {
                return [arr](auto G, auto J) {
                    constexpr int g = decltype(G)::value, j = decltype(J)::value;
                    return arr[g >> 2][(g & 3) * 4 + j];
                };
            };
            // One layer of the trunk / colour branch (save_idx < 0: nothing recorded here)
            auto layer = [&](auto F0c, auto NTc, auto KGc, auto& out, auto&& bget, const float* bias_l, int save_idx) {
                constexpr int F0 = decltype(F0c)::value, NT = decltype(NTc)::value, KG = decltype(KGc)::value;
                // no read-ahead past the end of the pass, nor past views_linears.0's four hidden tiles (the sigma tile's
                // fragments that follow them are walked, not read)
                constexpr bool LAST = (F0 + NT * KG == kUsedFrags) || F0 == layer_f0(8) || F0 == layer_f0(10);
                static_assert(layer_f0(5) % kSliceFrags == 0 && layer_f0(6) % kSliceFrags == 0, "input prefetch hooks sit on slice boundaries");
                // the next tile's point inputs: loaded after pts_linears.5's first slice opens, touched one layer later
                auto hook = [&]() {
                    if constexpr (MODE != kModeX && F0 == layer_f0(5)) load_point<MODE>(a, tile + gridDim.x, wave, m, nxt);
                    if constexpr (MODE != kModeX && F0 == layer_f0(6)) touch_point(nxt);
                };
                DIAG_BEGIN(dg);
                load_bias<NT>(out, bias_l);
                DIAG_END(dg, kDgBoundary);
                if (save_idx >= 0) {  // hidden layer: ReLU + record, in the MFMA shadow
                    const SaveSide<NT, KG / 2, 32 * NT> side{mk, &out[0], rows_rsrc(a.acts + (long)act_off(save_idx) * a.p_pad + p0 * (32 * NT), 32 * NT),
                                                             (uint32_t)((h * (32 * NT) + m) * 4), stage, rb, raddr, m, h};
                    run_layer<F0, NT, KG, LAST>(out, bget, ws, fr, side, hook);
                    side.finish();
                    store_mask(save_idx - kActA1);   // a1..a8 -> 0..7, v2 / v3 -> 9 / 10
                } else {
                    run_layer<F0, NT, KG, LAST>(out, bget, ws, fr, NoSide{}, hook);
                }
            };
            // ---- pts_linears.0 : PE(64) -> 256
            layer(ic<layer_f0(0)>{}, ic<8>{}, ic<8>{}, A, pe_get, bias_h + bias_off(0), kActA1 + 0);
            // ---- pts_linears.1..4 : 256 -> 256   (A -> B -> A -> B -> A)
#pragma unroll 1
            for (int l = 1; l <= 3; l += 2) {
                layer(ic<layer_f0(1)>{}, ic<8>{}, ic<32>{}, B, tiles_get(A), bias_h + l * 256, kActA1 + l);
                layer(ic<layer_f0(2)>{}, ic<8>{}, ic<32>{}, A, tiles_get(B), bias_h + (l + 1) * 256, kActA1 + l + 1);
            }
            // ---- pts_linears.5 : [PE(64) | 256] -> 256   (skip connection, face_nerf.py:61-62)
            layer(ic<layer_f0(5)>{}, ic<8>{}, ic<40>{}, B,
                  [&](auto G, auto J) {
                      constexpr int g = decltype(G)::value, j = decltype(J)::value;
                      if constexpr (g < 8) return pe[g][j];
                      else return A[(g - 8) >> 2][((g - 8) & 3) * 4 + j];
                  },
                  bias_h + bias_off(5), kActA1 + 5);
            // ---- pts_linears.6, .7
            layer(ic<layer_f0(6)>{}, ic<8>{}, ic<32>{}, A, tiles_get(B), bias_h + bias_off(6), kActA1 + 6);
            layer(ic<layer_f0(7)>{}, ic<8>{}, ic<32>{}, B, tiles_get(A), bias_h + bias_off(7), kActA1 + 7);
            // ---- views_linears.0 : [256 | dirPE(32)] -> 128.  The stream still carries alpha_linear as a fifth tile
            //      (the bf16 kernels use it); here its 36 fragments are walked without being read, and sigma is a
            //      256-term dot product on the vector unit: 128 FMAs per lane against 144 MFMAs (one row of 32 used).
            f32x16(&V4a)[4] = reinterpret_cast<f32x16(&)[4]>(V);
            layer(ic<layer_f0(8)>{}, ic<4>{}, ic<36>{}, V4a,
                  [&](auto G, auto J) {
                      constexpr int g = decltype(G)::value, j = decltype(J)::value;
                      if constexpr (g < 32) return B[g >> 2][(g & 3) * 4 + j];
                      else return pd[g - 32][j];
                  },
                  bias_h + bias_off(8), -1);
            {
                constexpr int f_from = layer_f0(8) + 4 * 36, f_to = layer_f0(9);
                static_assert(f_from / kSliceFrags == (f_to - 1) / kSliceFrags && f_to % kSliceFrags != 0, "the walk stays inside one open slice");
                static_for<(f_to - f_from) / 2>([&](auto I) { ws.template step_piece<f_from + 2 * decltype(I)::value>(); });
                fr.pref0 = fr.template issue<f_to>();
                fr.pref1 = fr.template issue<f_to + 1>();
            }
            {
                // B holds h7 (post-ReLU; its last tile was finished inside the layer above): this lane has the
                // channels 32t + 8q + 4h + i, the weights sit in LDS in the same order as a bias row
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                static_for<8>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    static_for<4>([&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        const f32x4 w = *reinterpret_cast<const f32x4*>(bias_h + kAlphaOff + 32 * t + 8 * q);
                        s0 = fmaf(w.x, B[t][4 * q + 0], s0);
                        s1 = fmaf(w.y, B[t][4 * q + 1], s1);
                        s2 = fmaf(w.z, B[t][4 * q + 2], s2);
                        s3 = fmaf(w.w, B[t][4 * q + 3], s3);
                    });
                });
                const float part = (s0 + s1) + (s2 + s3);
                sigma = bias_s[bias_off(8) + kSigmaChannel] + (part + __shfl_xor(part, 32, 64));   // both lane halves
            }
            {
                DIAG_BEGIN(dg);
                f32x16(&V4)[4] = reinterpret_cast<f32x16(&)[4]>(V);
                static_for<4>([&](auto T) { collect_signs<decltype(T)::value>(V4[decltype(T)::value], mk); });
                store_mask(kActV1 - kActA1);   // v1 -> 8
                relu_tiles<4>(V4);
                const SaveSide<4, 1, 128> side{mk, &V4[0], rows_rsrc(a.acts + (long)act_off(kActV1) * a.p_pad + p0 * 128, 128),
                                               (uint32_t)((h * 128 + m) * 4), stage, rb, raddr, m, h};
                static_for<4>([&](auto T) { side.flush_tile(T); });
                DIAG_END(dg, kDgBoundary);
            }
            // ---- views_linears.1, .2 : 128 -> 128   (V -> A[0..3] -> V[0..3])
            f32x16(&A4)[4] = reinterpret_cast<f32x16(&)[4]>(A);
            f32x16(&V4b)[4] = reinterpret_cast<f32x16(&)[4]>(V);
            layer(ic<layer_f0(9)>{}, ic<4>{}, ic<16>{}, A4, tiles_get(V), bias_h + bias_off(9), kActV1 + 1);
            layer(ic<layer_f0(10)>{}, ic<4>{}, ic<16>{}, V4b, tiles_get(A), bias_h + bias_off(10), kActV1 + 2);
            // ---- rgb_linear : 128 -> 3.  Like sigma: three 128-term dot products on the vector unit instead of a
            //      32-row MFMA tile of which three rows are used; its 16 fragments are walked with the padding.
            //      One tile at a time, fenced: left alone the scheduler hoists all 48 weight reads (192 registers).
            finish_pass<layer_f0(11)>(ws);
            rgb[0] = bias_s[bias_off(11) + 0];
            rgb[1] = bias_s[bias_off(11) + 1];
            rgb[2] = bias_s[bias_off(11) + 2];
            {
                float part[3] = {0.f, 0.f, 0.f};
                static_for<4>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    f32x16 vt = V[t];
                    asm volatile("" : "+v"(vt));   // the tile's 16 values in VGPRs before its weights are read
                    static_for<3>([&](auto Cc) {
                        constexpr int c = decltype(Cc)::value;
                        static_for<4>([&](auto Q) {
                            constexpr int q = decltype(Q)::value;
                            const f32x4 w = *reinterpret_cast<const f32x4*>(bias_h + kRgbOff + 128 * c + 32 * t + 8 * q);
                            part[c] = fmaf(w.x, vt[4 * q + 0], part[c]);
                            part[c] = fmaf(w.y, vt[4 * q + 1], part[c]);
                            part[c] = fmaf(w.z, vt[4 * q + 2], part[c]);
                            part[c] = fmaf(w.w, vt[4 * q + 3], part[c]);
                        });
                    });
                    asm volatile("" : "+v"(part[0]), "+v"(part[1]), "+v"(part[2])::"memory");
                });
                static_for<3>([&](auto Cc) {
                    constexpr int c = decltype(Cc)::value;
                    rgb[c] += part[c] + __shfl_xor(part[c], 32, 64);
                });
            }
        }

        DIAG_BEGIN(dg);
        if (valid && h == 0) {
            f32x4 o;
            o.x = rgb[0];
            o.y = rgb[1];
            o.z = rgb[2];
            o.w = sigma;
            *reinterpret_cast<f32x4*>(a.raw + P * 4) = o;
        }
        DIAG_END(dg, kDgStore);
        DIAG_ONLY(dg.acc[kDgTotal] += clock64() - t_tile;)
        cur = nxt;
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
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModeRays, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModeX, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModePts, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f32_kernel<kModeRays, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLdsTrain));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ntiles = (n_points + 127) / 128;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    MlpArgs a{packed, folded, x, rays, z, pts, dirs, (long)n_points, n_samples, raw, acts, (long)p_pad};
    ProfScope prof(s, n_points, acts ? IDN_PROF_MLP_FWD_SAVE : IDN_PROF_MLP_FWD);
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
