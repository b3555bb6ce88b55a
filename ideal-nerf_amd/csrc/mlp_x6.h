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
// TWO traps, both met in round 3.  (1) The hazard recogniser does not look inside asm: a piece word written by this
// instruction and read as an MFMA operand two (independent, back-to-back) MFMAs later came back as garbage.  The kernels
// therefore keep every conversion at least three DEPENDENT MFMAs away from the MFMA that reads its result (K-major forward:
// side work after the third MFMA of a step, a scheduling fence at the end of every step, the pieces are read from the next
// k-step on), and tools/audit_asm_loads.py rejects an asm vector result read by an MFMA within 8 instructions.  (2) As a
// compiler-visible conversion (__builtin_convertvector lowers to the same instruction) hipcc hoists the conversions of later
// steps forward and the training forward spills 97 registers to scratch (4.45 -> 5.0 ms).
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
// Two slots (96 KiB) everywhere but in the inference forward, which has the LDS for three (144 KiB): its pieces are then
// fetched two slices ahead and have more than a slice and a half to land before the barrier that opens their slice.
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

}  // namespace x6
}  // namespace idn
