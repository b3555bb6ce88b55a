// Arithmetic shared by the six-piece bf16 kernels (mlp_bf16x6.hip: forward; mlp_f32_bwd.hip: the delta chain):
// an fp32 value as the exact sum of three bf16 pieces, accumulator tiles <-> piece operands, the counted wait of
// a fragment triple.  gfx950 only.
#pragma once
#include "mlp_common.h"

namespace idn {
namespace x6 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// (low half, high half) = (bf16(x0), bf16(x1)), round to nearest even: one v_cvt_pk_bf16_f32, as inline asm.
// TWO traps, both met in round 3.  (1) A vector write followed by an MFMA read of the same register needs two wait states
// (tools/valu_mfma_hazard_ubench.hip: with none or one the MFMA reads the OLD value), and the hazard recogniser does not
// look inside asm: a piece word written here, `s_nop 0`, then the MFMA reading it -> garbage.  The K-major kernels keep every
// conversion three dependent MFMAs away from the MFMA that reads its result (side work after the third MFMA of a step, a
// scheduling fence at the end of every step, the pieces are read from the next k-step on; pieces made outside the pipeline
// pass through settle()), and tools/audit_asm_loads.py rejects an asm vector result read by an MFMA fewer than two
// instructions later.  (2) As a compiler-visible conversion (__builtin_convertvector lowers to the same instruction) hipcc
// hoists the conversions of later steps forward and the training forward spills 97 registers to scratch (4.45 -> 5.0 ms).
__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
    return r;
}
// one packed word of each of the three pieces of two fp32 values (inputs are ordinary VALU results)
__device__ __forceinline__ void split3(float x0, float x1, float& w1, float& w2, float& w3) {
    const unsigned p1 = cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);   // exact
    const unsigned p2 = cvt_pk_bf16(r0, r1);
    r0 = r0 - __uint_as_float(p2 << 16);
    r1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    w1 = __uint_as_float(p1);
    w2 = __uint_as_float(p2);
    w3 = __uint_as_float(cvt_pk_bf16(r0, r1));
}

// The three pieces of one 16-channel k-step of a wave's 32 points: the B operands of that k-step's six products.
struct KP {
    f32x4 p[3];
};
// Pieces produced OUTSIDE the k-step pipeline (the encodings at the start of a pass, the d rgb / d sigma inputs of the delta
// chain) and read by MFMAs soon after: tie them through a few wait states, so that no asm conversion result reaches an MFMA
// within the window the hazard recogniser would have covered had it seen the conversion (cvt_pk_bf16 above).
__device__ __forceinline__ void settle(f32x4 (&p)[3]) {
    asm volatile("s_nop 7\n\ts_nop 3" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]));
}
__device__ __forceinline__ void settle(KP& k) { settle(k.p); }
// The pieces of one 32-channel tile of activations: piece q, k-step s (two 16-channel k-steps per tile).
struct PTile6 {
    f32x4 p[3][2];
};
// accumulator tile -> pieces.  Word W (0..7) = registers 2W, 2W+1 -> word W & 3 of k-step W >> 2 (element j of
// k-step s in lane half h is channel 16 s + (j & 3) + 8 (j >> 2) + 4 h: how pack_bf16x6_kernel orders the weights).
template <bool RELU>
__device__ __forceinline__ void convert_tile(const f32x16& acc, PTile6& out) {
    static_for<8>([&](auto W_) {
        constexpr int w = decltype(W_)::value;
        float x0 = acc[2 * w], x1 = acc[2 * w + 1];
        if constexpr (RELU) {
            x0 = relu1(x0);
            x1 = relu1(x1);
        }
        float w1, w2, w3;
        split3(x0, x1, w1, w2, w3);
        out.p[0][w >> 2][w & 3] = w1;
        out.p[1][w >> 2][w & 3] = w2;
        out.p[2][w >> 2][w & 3] = w3;
    });
}

// The six-piece streams: 48-fragment ring slots (16 k-steps of three fragments), 12 LDS-DMA pieces per wave and slice.
// The forward kernels have the LDS for three slots (144 KiB): pieces are fetched two slices ahead and have more than a slice
// and a half to land before the barrier that opens their slice -- and so have the row stores of the training variant, which
// retire in issue order with them.  The delta chain keeps two (96 KiB).
using WStream6 = WStreamT<4, kX6SliceFrags>;
using WStream6x3 = WStreamT<4, kX6SliceFrags, 3>;
static_assert(WStream6::kPieces == 12 && kX6NumSlices % 3 == 0, "two pieces at each of the first six k-steps of a slice; slot phase static across passes");
// fragment Q of the k-step at stream fragment F (F a multiple of 3)
template <int F, class WS = WStream6>
__device__ __forceinline__ f32x4 issue6(const FragReader& fr) {
    return fr.template issue<F, WS::kSlotsT * kX6SliceFrags>();
}
// the pieces (if any) issued at the k-step that starts at fragment F: two per k-step, so a wave's twelve pieces of the
// next slice are all issued in the first six k-steps of a slice (before anything the training variants store in its
// second half: those stores stay YOUNGER than the pieces)
template <int F, class WS>
__device__ __forceinline__ void step_pieces6(WS& ws) {
    static_assert(F % kX6KFrags == 0, "k-steps");
    constexpr int ks = (F % kX6SliceFrags) / kX6KFrags;
    ws.template step_piece_at<F, 2 * ks>();
    ws.template step_piece_at<F, 2 * ks + 1>();
}
// End of a pass: walk the unused tail of the stream k-step by k-step without reading it (finish_pass of mlp_common.h).
template <int F_END, int STREAM_FRAGS, class WS>
__device__ __forceinline__ void finish_pass6(WS& ws) {
    static_assert(F_END % kX6KFrags == 0 && STREAM_FRAGS % kX6SliceFrags == 0, "k-steps; whole slices");
    static_for<(STREAM_FRAGS - F_END) / kX6KFrags>([&](auto I) {
        constexpr int f = F_END + kX6KFrags * decltype(I)::value;
        if constexpr (f % kX6SliceFrags == 0) ws.open_slice();
        step_pieces6<f>(ws);
    });
}
template <class WS>
constexpr int mlp_lds6() { return WS::kSlotsT * kX6SliceFrags * kFragBytes + kBiasFloats * 4; }
constexpr int kMlpLds6 = mlp_lds6<WStream6>();

// all but the newest `Newer` LDS reads of this wave have completed => the three fragments are valid
template <int Newer>
__device__ __forceinline__ void retire3(f32x4 (&v)[3]) {
    if constexpr (Newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])::"memory");
}


// ---------------------------------------------------------------------------
// K-major layers: shared by the forward (mlp_bf16x6.hip) and the delta chain (mlp_f32_bwd.hip)
// ---------------------------------------------------------------------------
// tile-step at which word W (of 4) of the next k-step's pieces is produced: after tile 0 (whose accumulator the first
// k-step of the NEXT layer reads: it is complete after step (KS - 1, 0)), spread over the k-step
constexpr int conv_slot(int nt, int w) { return nt == 1 ? 0 : 1 + (w * (nt - 1)) / 4; }

// tile-steps at which the two row stores of a prepared half tile are issued (training): behind the words of their quad, and
// in an 8-tile layer at tile-steps 6 and 7, i.e. AFTER the twelve LDS-DMA pieces a wave issues in steps 0..5 of a slice
constexpr int store_slot(int nt, int q) { return nt >= 8 ? nt - 2 + q : conv_slot(nt, 3); }

// Row stores issued by step (s, t) of a layer whose k-steps [from, to) each prepare a half tile (training; `to` may be KS + 1:
// the layer's last k-step prepares the next layer's first).  On gfx9 stores count in vmcnt, vmcnt retires in issue order, and
// a slice barrier waits for the wave's pieces of the next slice -- which are issued in steps 0..5 of a slice.  The stores
// issued from step 5 of a slice on are therefore YOUNGER than those pieces and may stay in flight across the barrier: the
// barrier waits vmcnt(their number) instead of vmcnt(0).  The number is counted from this table at compile time -- never
// more than were issued (tools/audit_asm_loads.py checks the compiled ISA).
template <int NT, int FROM, int TO, bool SAVE>
struct Stores {
    static constexpr int at(int s, int t) {
#ifdef IDN_TIMING_NO_ROW_STORES   // the timing-only build issues no row stores: nothing may be counted as in flight
        return 0;
#endif
        return (SAVE && s + 1 >= FROM && s + 1 < TO) ? (t == store_slot(NT, 0)) + (t == store_slot(NT, 1)) : 0;
    }
    // stores of the steps [i0, i1] of the layer (clipped to it)
    static constexpr int in_steps(int i0, int i1, int np) {
        int n = 0;
        for (int i = (i0 < 0 ? 0 : i0); i <= i1 && i < np; ++i) n += at(i / NT, i % NT);
        return n;
    }
};
constexpr int kFirstYoungStep = 5;   // in-slice step from which a wave's stores are younger than the pieces it issued in that slice
struct NoStores {   // the layer before stores nothing (or nothing is known about it: counting fewer stores than were issued is always safe)
    static constexpr int at(int, int) { return 0; }
    static constexpr int in_steps(int, int, int) { return 0; }
};
// The row stores a wave may leave in flight at the barrier that follows step `pi` (-1: the barrier in front of the layer's
// first step) of a layer with NP steps and stores ST, behind a layer with PNP steps and stores PST.  The barrier opens the
// slice whose pieces were issued in steps 0..5 of the slice AHEAD slices back (two-slot ring: the slice that ends here;
// three-slot ring: the one before it), so every store since step kFirstYoungStep of THAT slice is younger than those
// pieces.  Stores further back than the layer before are not counted (a stricter wait, never a weaker one).
template <class ST, int NP, class PST, int PNP, int AHEAD>
constexpr int young_stores(int pi) {
    constexpr int kSliceSteps = kX6SliceFrags / kX6KFrags;
    const int first = pi - (AHEAD * kSliceSteps - 1) + kFirstYoungStep;   // first step of the window [first, pi]
    return ST::in_steps(first, pi, NP) + (first < 0 ? PST::in_steps(PNP + first, PNP - 1, PNP) : 0);
}

// One layer (forward) or stage (delta chain), K-major.  LAST: nothing is read ahead past its last step (the end of the
// stream, or padding that is walked, not read).  On entry O[0..NT) hold the layer's biases (the stage's zeros), B the pieces of its k-step 0, `pref` the fragments
// of its first step (unless the layer starts on a slice boundary).  side(ic<s>, ic<t>, Bn) runs inside step (s, t) and
// fills Bn, the pieces of the NEXT k-step (of this layer, or -- in the layer's last k-step -- of the next layer, along
// with that layer's biases).  ST: the row stores the side work issues (training), see Stores; PST, PNP: those of the layer
// before and its number of steps.
template <int F0, int NT, int KS, bool LAST, class PST, int PNP, class ST, class Side, class Hook, class WS>
__device__ __forceinline__ void run_layer(f32x16* O, KP& B, Side&& side, WS& ws, FragReader& fr, f32x4 (&pref)[3], Hook&& after_open) {
    constexpr int NP = NT * KS;
    static_assert(F0 % kX6KFrags == 0, "triples");
    if constexpr (F0 % kX6SliceFrags == 0) {
        ws.template open_slice<young_stores<ST, NP, PST, PNP, WS::kAheadT>(-1)>();   // (training: the young row stores of the layer before stay in flight)
        after_open();
        static_for<3>([&](auto Q) { pref[decltype(Q)::value] = issue6<F0 + decltype(Q)::value, WS>(fr); });
        retire3<0>(pref);
    } else {
        static_assert(std::is_same_v<std::decay_t<Hook>, NoHook>, "a hook needs a layer that starts on a slice boundary");
    }
    f32x4 a[3] = {pref[0], pref[1], pref[2]};
    KP Bn = B;
    static_for<NP>([&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        constexpr int s = pi / NT, t = pi % NT;
        constexpr int f = F0 + kX6KFrags * pi;
        constexpr bool next_crosses = ((f + kX6KFrags) % kX6SliceFrags == 0);
        constexpr bool has_next = !(LAST && pi + 1 == NP);
        f32x4 n[3] = {a[0], a[1], a[2]};
        if constexpr (!next_crosses && has_next) {
            static_for<3>([&](auto Q) { n[decltype(Q)::value] = issue6<f + kX6KFrags + decltype(Q)::value, WS>(fr); });
            if constexpr (pi > 0) retire3<3>(a);   // (step 0's arrived retired)
        } else {
            if constexpr (pi > 0) retire3<0>(a);
        }
        step_pieces6<f>(ws);
        // (an order of the six products that changes fewer MFMA operands between consecutive instructions -- w3a1, w1a1, w2a1,
        //  w2a2, w1a2, w1a3 and the reverse in odd tile-steps -- measured -0.5 %: profiles/r03_ab_x6_product_order.log)
        O[t] = mfma_bf(a[0], B.p[0], O[t]);   // w1 a1
        O[t] = mfma_bf(a[0], B.p[1], O[t]);   // w1 a2
        O[t] = mfma_bf(a[1], B.p[0], O[t]);   // w2 a1
        side(ic<s>{}, ic<t>{}, Bn);           // the next k-step's pieces, in this one's MFMA shadow
        O[t] = mfma_bf(a[1], B.p[1], O[t]);   // w2 a2
        O[t] = mfma_bf(a[0], B.p[2], O[t]);   // w1 a3
        O[t] = mfma_bf(a[2], B.p[0], O[t]);   // w3 a1
        if constexpr (next_crosses && pi + 1 < NP) {
            ws.template open_slice<young_stores<ST, NP, PST, PNP, WS::kAheadT>(pi)>();
            static_for<3>([&](auto Q) { n[decltype(Q)::value] = issue6<f + kX6KFrags + decltype(Q)::value, WS>(fr); });
        }
        a[0] = n[0];
        a[1] = n[1];
        a[2] = n[2];
        if constexpr (t == NT - 1) B = Bn;
        // a step's side work stays in its step (mlp_x6.h, cvt_pk_bf16: the asm conversions must keep their distance from the MFMAs
        // that read their results, which are those of the NEXT k-step)
        __builtin_amdgcn_sched_barrier(0);
    });
    // hand over retired fragments (those of the next layer's first step, when this layer ends inside a slice)
    if constexpr (!LAST && (F0 + kX6KFrags * NP) % kX6SliceFrags != 0) retire3<0>(a);
    pref[0] = a[0];
    pref[1] = a[1];
    pref[2] = a[2];
}

}  // namespace x6
}  // namespace idn
