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
}  // namespace idn
