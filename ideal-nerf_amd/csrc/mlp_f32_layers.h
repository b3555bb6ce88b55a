// The fp32-MFMA layer loop shared by the fused PE + MLP forward (mlp_f32.hip) and the fused ray kernel
// (render_fused.hip).  gfx950 only.
#pragma once
#include "mlp_common.h"

namespace idn {

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// One layer: out[t] += W[32t.., :] . B  for t = 0..NT-1, tile after tile (the stream is
// tile-major), each tile's KG k-groups consumed as KG/2 fragment pairs = 8 MFMAs per step.
//   * the reads of pair p+1 are issued ahead of the MFMAs of pair p;
//   * one prefetch piece of the next slice is issued per step in the first half of a slice;
//   * side(t, s, half) is called after each group of four MFMAs: VALU / LDS work placed
//     there issues while the matrix pipe is busy (finished tiles' ReLU, the next tile's bias).
// F0 = index of the layer's first fragment in the stream.  On entry fr.pref* hold pair F0
// in flight unless F0 opens a slice; on exit they hold pair F0 + NT*KG likewise.
// LAST: the final layer of a pass must not prefetch the pair after its own last one -- nobody
// would retire that read, the compiler would treat its destination registers as dead and
// reuse them (e.g. as a global address) while the LDS data is still on its way
// (tools/audit_asm_loads.py checks the ISA for this).
template <int F0, int NT, int KG, bool LAST = false, class BGet, class Side, class Hook = NoHook, class WS = WStream>
__device__ __forceinline__ void run_layer(f32x16 (&out)[NT], BGet&& bget, WS& ws, FragReader& fr, Side&& side,
                                          Hook&& after_open = NoHook{}) {
    constexpr int STEPS = KG / 2, NP = NT * STEPS;
    static_assert(KG % 2 == 0 && F0 % 2 == 0, "fragments are consumed in pairs");
    if constexpr (F0 % kSliceFrags == 0) {
        ws.open_slice();
        after_open();   // global loads issued here have a whole slice to land before the next barrier's vmcnt(0)
        fr.pref0 = fr.template issue<F0>();
        fr.pref1 = fr.template issue<F0 + 1>();
    }
    f32x4 a0 = fr.pref0, a1 = fr.pref1;
    static_for<NP>([&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        constexpr int t = pi / STEPS, s = pi % STEPS, g0 = 2 * s, g1 = g0 + 1;
        constexpr int f = F0 + 2 * pi;
        constexpr bool next_crosses = ((f + 2) % kSliceFrags == 0);
        f32x4 n0 = a0, n1 = a1;
        if constexpr (!next_crosses && !(LAST && pi + 1 == NP)) {
            n0 = fr.template issue<f + 2>();
            n1 = fr.template issue<f + 3>();
            FragReader::retire<2>(a0, a1);
        } else {
            FragReader::retire<0>(a0, a1);
        }
        ws.template step_piece<f>();
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

// One INFERENCE pass of the fp32 FaceNeRF on this wave's 32 points (mlp_f32_kernel<MODE, false> and the fused ray kernel):
// pe / pd = this lane's half of the 64 point features and 32 direction features, the folded bias block in LDS at bias_s
// (bias_h = bias_s + 4 h), the weights through the ring.  Two sets of eight 32 x 32 tiles take turns as a layer's input (B
// operands) and output (accumulators); a finished layer's output is ReLU'd in place -- everything but tile 0's bias in MFMA
// shadows (LayerSide), the last tile's ReLU owed to the next layer -- and read by the next.  after_open5 / after_open6 run
// right after the first slice of pts_linears.5 / .6 opens (the next pass's point inputs are loaded there and touched one
// layer later: a whole slice to land before the next barrier's vmcnt(0)).  -> raw rgb (3) and sigma of the lane's point.
template <class WS, class Hook5, class Hook6>
__device__ __forceinline__ void f32_inference_pass(const float (&pe)[8][4], const float (&pd)[4][4], const float* bias_s, const float* bias_h,
                                                   WS& ws, FragReader& fr, Hook5&& after_open5, Hook6&& after_open6, float (&rgb)[3],
                                                   float& sigma) {
    f32x16 A[8], B[8], V[5];
    auto pe_get = [&](auto G, auto J) { return pe[decltype(G)::value][decltype(J)::value]; };
    auto tiles_get = [](f32x16* arr) {
        return [arr](auto G, auto J) {
            constexpr int g = decltype(G)::value, j = decltype(J)::value;
            return arr[g >> 2][(g & 3) * 4 + j];
        };
    };
    auto layer = [&](auto F0c, auto NTc, auto KGc, auto DEFERc, auto& out, f32x16* deferred, auto&& bget, const float* bias_l) {
        constexpr int F0 = decltype(F0c)::value, NT = decltype(NTc)::value, KG = decltype(KGc)::value;
        // no read-ahead past the end of the pass, nor past views_linears.0's four hidden tiles (the sigma tile's
        // fragments that follow them are walked, not read)
        constexpr bool LAST = (F0 + NT * KG == kUsedFrags) || F0 == layer_f0(8) || F0 == layer_f0(10);
        constexpr bool DEFER = decltype(DEFERc)::value != 0;
        static_assert(layer_f0(5) % kSliceFrags == 0 && layer_f0(6) % kSliceFrags == 0, "input prefetch hooks sit on slice boundaries");
        auto hook = [&]() {
            if constexpr (F0 == layer_f0(5)) after_open5();
            if constexpr (F0 == layer_f0(6)) after_open6();
        };
        bias_tile(out[0], bias_l);
        run_layer<F0, NT, KG, LAST>(out, bget, ws, fr, LayerSide<NT, KG / 2, DEFER>{&out[0], deferred, bias_l}, hook);
    };
    // ---- pts_linears.0 : PE(64) -> 256
    layer(ic<layer_f0(0)>{}, ic<8>{}, ic<8>{}, ic<0>{}, A, nullptr, pe_get, bias_h + bias_off(0));
    // ---- pts_linears.1..4 : 256 -> 256   (A -> B -> A -> B -> A)
#pragma unroll 1
    for (int l = 1; l <= 3; l += 2) {
        layer(ic<layer_f0(1)>{}, ic<8>{}, ic<32>{}, ic<1>{}, B, &A[7], tiles_get(A), bias_h + l * 256);
        layer(ic<layer_f0(2)>{}, ic<8>{}, ic<32>{}, ic<1>{}, A, &B[7], tiles_get(B), bias_h + (l + 1) * 256);
    }
    // ---- pts_linears.5 : [PE(64) | 256] -> 256   (skip connection, face_nerf.py:61-62)
    layer(ic<layer_f0(5)>{}, ic<8>{}, ic<40>{}, ic<1>{}, B, &A[7],
          [&](auto G, auto J) {
              constexpr int g = decltype(G)::value, j = decltype(J)::value;
              if constexpr (g < 8) return pe[g][j];
              else return A[(g - 8) >> 2][((g - 8) & 3) * 4 + j];
          },
          bias_h + bias_off(5));
    // ---- pts_linears.6, .7
    layer(ic<layer_f0(6)>{}, ic<8>{}, ic<32>{}, ic<1>{}, A, &B[7], tiles_get(B), bias_h + bias_off(6));
    layer(ic<layer_f0(7)>{}, ic<8>{}, ic<32>{}, ic<1>{}, B, &A[7], tiles_get(A), bias_h + bias_off(7));
    // ---- views_linears.0 : [256 | dirPE(32)] -> 128.  The stream still carries alpha_linear as a fifth tile
    //      (the bf16 kernels use it); here its 36 fragments are walked without being read, and sigma is a
    //      256-term dot product on the vector unit: 128 FMAs per lane against 144 MFMAs (one row of 32 used).
    f32x16(&V4a)[4] = reinterpret_cast<f32x16(&)[4]>(V);
    layer(ic<layer_f0(8)>{}, ic<4>{}, ic<36>{}, ic<1>{}, V4a, &B[7],
          [&](auto G, auto J) {
              constexpr int g = decltype(G)::value, j = decltype(J)::value;
              if constexpr (g < 32) return B[g >> 2][(g & 3) * 4 + j];
              else return pd[g - 32][j];
          },
          bias_h + bias_off(8));
    relu_regs<0, 16>(V[3]);   // the sigma tile used to give the fourth tile's ReLU its shadow
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
    // ---- views_linears.1, .2 : 128 -> 128   (V -> A[0..3] -> V[0..3])
    f32x16(&A4)[4] = reinterpret_cast<f32x16(&)[4]>(A);
    f32x16(&V4b)[4] = reinterpret_cast<f32x16(&)[4]>(V);
    layer(ic<layer_f0(9)>{}, ic<4>{}, ic<16>{}, ic<0>{}, A4, nullptr, tiles_get(V), bias_h + bias_off(9));
    layer(ic<layer_f0(10)>{}, ic<4>{}, ic<16>{}, ic<1>{}, V4b, &A[3], tiles_get(A), bias_h + bias_off(10));
    // ---- rgb_linear : 128 -> 3.  Like sigma: three 128-term dot products on the vector unit instead of a
    //      32-row MFMA tile of which three rows are used; its 16 fragments are walked with the padding.
    //      One tile at a time, fenced: left alone the scheduler hoists all 48 weight reads (192 registers).
    relu_regs<0, 16>(V[3]);   // owed by views_linears.2 (its last tile's ReLU was deferred)
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

}  // namespace idn
