// Fused positional-encoding + FaceNeRF MLP forward, bf16x3 arithmetic (gfx950).
//
// Same computation, stream geometry and tile-major layer driver as mlp_f32.hip, but each
// fp32 product a*b is evaluated on the bf16 matrix pipe as
//      a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,      x_hi = bf16(x), x_lo = bf16(x - x_hi)
// with fp32 accumulation (v_mfma_f32_32x32x16_bf16, 3 per 16 channels): ~2^-16 relative
// error per product, ~1.5e-5 on the network output (hard part 3) -- inside
// the 1e-4 RGB budget -- at 16/3 = 5.3x the fp32-MFMA rate, and on a pipe that does not share
// the vector ALUs (DESIGN.md "What bounds the fp32 kernel").
//
// A fragment pair (2p, 2p+1) of the stream is the hi and the lo half of one 16-channel
// k-step: 8 bf16 per lane each.  An accumulator tile (32 channels x 32 points, fp32) is
// ReLU'd and split into two k-steps of packed hi / lo B operands in registers; element j of
// k-step s in lane half h is register 8s+j = channel 16s + (j&3) + 8(j>>2) + 4h, which is how
// pack_bf16x3_kernel orders the weights.
#include <type_traits>

#include "mlp_common.h"

namespace idn {

// The x3 kernel exists in two 16-bit formats, compiled from this one source (mlp_fp16x3.hip defines
// IDN_X3_FP16 and includes it):
//   bf16 x3: 8+8 significand bits per operand -> ~1.5e-5 on the network output; no range limit.
//   fp16 x3: 11+11 bits -> ~5e-7, i.e. fp32-like parity at the same MFMA count, for |activations| < 6.5e4
//            (fp16's range; beyond it the split saturates: finite but wrong).
#ifdef IDN_X3_FP16
#define X3_NS x3_fp16
#define X3_KERNEL mlp_fp16x3_kernel
#define X3_LAUNCH launch_mlp_fp16x3
#else
#define X3_NS x3_bf16
#define X3_KERNEL mlp_bf16x3_kernel
#define X3_LAUNCH launch_mlp_bf16x3
#endif
namespace X3_NS {
#ifdef IDN_X3_FP16
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// one packed word of the hi part and of the lo part from two fp32 values.  hi is rounded toward zero
// (v_cvt_pkrtz_f16_f32, never overflows to inf); lo = x - hi is exact in fp32 and carries the next 11 bits.
__device__ __forceinline__ void split2(float x0, float x1, float& hi_w, float& lo_w) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi_w = __builtin_bit_cast(float, h);
    // x - float(h) in one instruction per value: v_fma_mix_f32 reads the f16 half directly (op_sel_hi marks
    // source 0 as f16, op_sel picks its half).  Inputs are ordinary VALU results, never raw MFMA outputs.
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi_w), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi_w), "v"(x1));
    lo_w = __builtin_bit_cast(float, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
#else
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// (lo half, hi half) = (bf16(x0), bf16(x1)), round to nearest even.  The two wait states behind it are part of the
// instruction as far as this kernel is concerned: an MFMA that reads a register a vector instruction wrote needs them
// (tools/valu_mfma_hazard_ubench.hip), the hazard recogniser does not see inside asm, and the plain-bf16 kernel had four
// places per pass where the conversion, `s_nop 0` and the MFMA reading its result followed each other -- the MFMA then
// took the register's OLD contents for two of its sixteen channels (found by tools/audit_asm_loads.py's fourth check in
// round 3; inside the plain-bf16 mode's 6.5e-3 it had gone unnoticed).
__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(x0), "v"(x1));
    return r;
}
// one packed word of the hi part and of the lo part from two fp32 values
__device__ __forceinline__ void split2(float x0, float x1, float& hi_w, float& lo_w) {
    const unsigned h = cvt_pk_bf16(x0, x1);
    hi_w = __uint_as_float(h);
    // (the two subtractions as one v_pk_add_f32: same-box A/B -0.4 %, left as two)
    const float f0 = __uint_as_float(h << 16), f1 = __uint_as_float(h & 0xffff0000u);
    lo_w = __uint_as_float(cvt_pk_bf16(x0 - f0, x1 - f1));
}
#endif
}  // namespace X3_NS
using namespace X3_NS;

// Packed activations of one 32-channel tile: two k-steps, hi and lo.
struct BTile {
    f32x4 hi[2], lo[2];
};
// word W (0..7) of a tile: registers 2W, 2W+1 of the accumulator -> word W&3 of k-step W>>2
template <int W, bool RELU>
__device__ __forceinline__ void convert_word(const f32x16& acc, BTile& out) {
    float x0 = acc[2 * W], x1 = acc[2 * W + 1];
    if constexpr (RELU) {
        x0 = relu1(x0);
        x1 = relu1(x1);
    }
    float hw, lw;
    split2(x0, x1, hw, lw);
    out.hi[W >> 2][W & 3] = hw;
    out.lo[W >> 2][W & 3] = lw;
}
template <int W0, int CNT, bool RELU>
__device__ __forceinline__ void convert_words(const f32x16& acc, BTile& out) {
    static_for<CNT>([&](auto I) {
        constexpr int w = W0 + decltype(I)::value;
        if constexpr (w < 8) convert_word<w, RELU>(acc, out);
    });
}

// In-shadow work of one layer (bf16 MFMAs leave the vector ALUs free):
//   * `pend` (the previous tile's finished accumulator) is ReLU'd + split into out[t-1];
//     for tile 0 with DEFER it is the previous LAYER's last tile, written into `deferred`;
//   * the bias of tile t+1 is loaded into the idle accumulator.
template <int NT, int STEPS, bool DEFER>
struct SideBf {
    BTile* out;
    BTile* deferred;
    f32x16* pend;
    f32x16* acc;             // acc[2]: tile t accumulates in acc[t & 1]
    const float* bias_half;  // bias_s + layer offset + 4h
    template <int T, int S, int H>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>, ic<H>) const {
        constexpr int C = (8 + STEPS - 1) / STEPS;       // words per step
        constexpr int CD = (16 + STEPS - 1) / STEPS;     // deferred tile: done by STEPS/2
        if constexpr (H == 0) {
            if constexpr (T > 0) convert_words<S * C, C, true>(*pend, out[T - 1]);
            if constexpr (T == 0 && DEFER) convert_words<S * CD, CD, true>(*pend, *deferred);
        } else {
            if constexpr (T + 1 < NT && S < 4) bias_quad<S>(acc[(T + 1) & 1], bias_half + 32 * (T + 1));
        }
    }
};

// One layer, tile-major: for each n-tile t, KS k-steps of (A_hi, A_lo) x (B_hi, B_lo).
// On exit `pend` holds the last tile's accumulator (to be converted by the caller / next layer).
template <int F0, int NT, int KS, bool DEFER, class BHi, class BLo, class Hook = NoHook>
__device__ __forceinline__ void run_layer_bf(BTile* out, BTile* deferred, f32x16& pend, f32x16 (&acc)[2],
                                             const float* bias_half, BHi&& bhi, BLo&& blo, WStream& ws, FragReader& fr,
                                             Hook&& after_open = NoHook{}) {
    constexpr int NP = NT * KS;
    // the last layer of a pass must not prefetch past itself (see run_layer in mlp_f32.hip)
    constexpr bool LAST = (F0 + 2 * NP == kUsedFrags);
    static_assert(F0 % 2 == 0, "fragments are consumed in (hi, lo) pairs");
    bias_tile(acc[0], bias_half);
    const SideBf<NT, KS, DEFER> side{out, deferred, &pend, &acc[0], bias_half};
    if constexpr (F0 % kSliceFrags == 0) {
        ws.open_slice();
        after_open();
        fr.pref0 = fr.template issue<F0>();
        fr.pref1 = fr.template issue<F0 + 1>();
    } else {
        static_assert(std::is_same_v<std::decay_t<Hook>, NoHook>, "a hook needs a layer that starts on a slice boundary");
    }
    f32x4 a0 = fr.pref0, a1 = fr.pref1;
    static_for<NP>([&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        constexpr int t = pi / KS, s = pi % KS;
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
        const f32x4 bh = bhi(ic<s>{}), bl = blo(ic<s>{});
        acc[t & 1] = mfma_bf(a0, bh, acc[t & 1]);   // hi * hi
        side(ic<t>{}, ic<s>{}, ic<0>{});
        acc[t & 1] = mfma_bf(a0, bl, acc[t & 1]);   // hi * lo
        side(ic<t>{}, ic<s>{}, ic<1>{});
        acc[t & 1] = mfma_bf(a1, bh, acc[t & 1]);   // lo * hi
        if constexpr (s == KS - 1) pend = acc[t & 1];  // tile t done: hand it to the converter
        if constexpr (next_crosses && pi + 1 < NP) {
            ws.open_slice();
            n0 = fr.template issue<f + 2>();
            n1 = fr.template issue<f + 3>();
        }
        a0 = n0;
        a1 = n1;
    });
    fr.pref0 = a0;
    fr.pref1 = a1;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void X3_KERNEL(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_s = reinterpret_cast<float*>(smem + kRingFrags * kFragBytes);

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

    PointIn cur, nxt;
    load_point<MODE>(a, blockIdx.x, wave, m, cur);
    nxt = cur;
#ifdef IDN_TIMING_PE_ONCE
    f32x4 pe_hi[4], pe_lo[4], pd_hi[2], pd_lo[2];
#endif
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long P = tile * 128 + wave * 32 + m;
        const bool valid = P < a.n_points;
        const long Pc = valid ? P : a.n_points - 1;

        // ---- inputs: this lane's half of the 64 point features and 32 direction features.
        // Element j of k-step s, lane half h = feature 16s + (j&3) + 8(j>>2) + 4h.
#ifdef IDN_TIMING_PE_ONCE   // timing-only (wrong results): the encoding of the block's first tile serves every pass
#else
        f32x4 pe_hi[4], pe_lo[4], pd_hi[2], pd_lo[2];
#endif
        auto pack_feats = [&](auto&& feat, f32x4* ohi, f32x4* olo, auto NKS) {
            static_for<decltype(NKS)::value>([&](auto S_) {
                constexpr int s = decltype(S_)::value;
                static_for<4>([&](auto W_) {
                    constexpr int w = decltype(W_)::value;
                    constexpr int j0 = 2 * w, j1 = 2 * w + 1;
                    constexpr int k0 = 16 * s + (j0 & 3) + 8 * (j0 >> 2), k1 = 16 * s + (j1 & 3) + 8 * (j1 >> 2);
                    float hw, lw;
                    split2(feat(ic<k0>{}), feat(ic<k1>{}), hw, lw);
                    ohi[s][w] = hw;
                    olo[s][w] = lw;
                });
            });
        };
#ifdef IDN_TIMING_PE_ONCE
        if (tile == blockIdx.x)
#endif
        input_features<MODE>(a, Pc, h, pln, cur, [&](auto&& fpt, auto&& fdir) {
            pack_feats(fpt, pe_hi, pe_lo, ic<4>{});
            pack_feats(fdir, pd_hi, pd_lo, ic<2>{});
        });

        BTile A[8], B[8], V[4];
        f32x16 acc[2], pend;
        auto tiles_hi = [](BTile* arr) { return [arr](auto S_) { constexpr int s = decltype(S_)::value; return arr[s >> 1].hi[s & 1]; }; };
        auto tiles_lo = [](BTile* arr) { return [arr](auto S_) { constexpr int s = decltype(S_)::value; return arr[s >> 1].lo[s & 1]; }; };
        auto pe_h = [&](auto S_) { return pe_hi[decltype(S_)::value]; };
        auto pe_l = [&](auto S_) { return pe_lo[decltype(S_)::value]; };

        // ---- pts_linears.0 : PE(64) -> 256
        run_layer_bf<layer_f0(0), 8, 4, false>(A, nullptr, pend, acc, bias_h + bias_off(0), pe_h, pe_l, ws, fr);
        // ---- pts_linears.1..4 (A -> B -> A -> B -> A); each layer first converts the previous layer's last tile
#pragma unroll 1
        for (int l = 1; l <= 3; l += 2) {
            run_layer_bf<layer_f0(1), 8, 16, true>(B, &A[7], pend, acc, bias_h + l * 256, tiles_hi(A), tiles_lo(A), ws, fr);
            run_layer_bf<layer_f0(2), 8, 16, true>(A, &B[7], pend, acc, bias_h + (l + 1) * 256, tiles_hi(B), tiles_lo(B), ws, fr);
        }
        // ---- pts_linears.5 : [PE(64) | 256] -> 256
        run_layer_bf<layer_f0(5), 8, 20, true>(
            B, &A[7], pend, acc, bias_h + bias_off(5),
            [&](auto S_) { constexpr int s = decltype(S_)::value; if constexpr (s < 4) return pe_hi[s]; else return A[(s - 4) >> 1].hi[(s - 4) & 1]; },
            [&](auto S_) { constexpr int s = decltype(S_)::value; if constexpr (s < 4) return pe_lo[s]; else return A[(s - 4) >> 1].lo[(s - 4) & 1]; },
            ws, fr, [&]() { load_point<MODE>(a, tile + gridDim.x, wave, m, nxt); });
        // ---- pts_linears.6, .7
        run_layer_bf<layer_f0(6), 8, 16, true>(A, &B[7], pend, acc, bias_h + bias_off(6), tiles_hi(B), tiles_lo(B), ws, fr,
                                               [&]() { touch_point(nxt); });
        run_layer_bf<layer_f0(7), 8, 16, true>(B, &A[7], pend, acc, bias_h + bias_off(7), tiles_hi(A), tiles_lo(A), ws, fr);
        // ---- views_linears.0 (+ alpha_linear as channel 128): [256 | dirPE(32)] -> 160.
        //      Tiles 0..3 are hidden units; tile 4 (last) is never converted: its row 0 is sigma.
        BTile Vx[5];
        run_layer_bf<layer_f0(8), 5, 18, true>(
            Vx, &B[7], pend, acc, bias_h + bias_off(8),
            [&](auto S_) { constexpr int s = decltype(S_)::value; if constexpr (s < 16) return B[s >> 1].hi[s & 1]; else return pd_hi[s - 16]; },
            [&](auto S_) { constexpr int s = decltype(S_)::value; if constexpr (s < 16) return B[s >> 1].lo[s & 1]; else return pd_lo[s - 16]; },
            ws, fr);
        const float sigma = pend[0];  // channel 128 = tile 4, register 0, lane half 0
        static_for<4>([&](auto T) { V[decltype(T)::value] = Vx[decltype(T)::value]; });
        // ---- views_linears.1, .2 : 128 -> 128 (V -> A[0..3] -> V)
        run_layer_bf<layer_f0(9), 4, 8, false>(A, nullptr, pend, acc, bias_h + bias_off(9), tiles_hi(V), tiles_lo(V), ws, fr);
        run_layer_bf<layer_f0(10), 4, 8, true>(V, &A[3], pend, acc, bias_h + bias_off(10), tiles_hi(A), tiles_lo(A), ws, fr);
        // ---- rgb_linear : 128 -> 3 (rows 0..2 of one tile)
        BTile none[1];
        run_layer_bf<layer_f0(11), 1, 8, true>(none, &V[3], pend, acc, bias_h + bias_off(11), tiles_hi(V), tiles_lo(V), ws, fr);
        finish_pass<kUsedFrags>(ws);

        if (valid && h == 0) {
            f32x4 o;
            o.x = pend[0];
            o.y = pend[1];
            o.z = pend[2];
            o.w = sigma;
            *reinterpret_cast<f32x4*>(a.raw + P * 4) = o;
        }
        cur = nxt;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}

#ifndef IDN_X3_FP16   // the plain-bf16 kernel is compiled once, with the bf16 x3 kernel
// ===========================================================================================
// Plain bf16 (IDN_PREC_BF16): one bf16 MFMA per 16 channels, weights and activations rounded to
// bf16 once, fp32 accumulate.  ~1e-2 relative on the raw output: reserved for
// BASELINE config 5, which is judged by PSNR.  Hi-only stream: half the fragments (plain_f0).
// A fragment pair = two consecutive k-steps.
// ===========================================================================================
struct PTile {
    f32x4 v[2];  // two k-steps of 8 packed bf16
};
template <int W, bool RELU>
__device__ __forceinline__ void convert_word_plain(const f32x16& acc, PTile& out) {
    float x0 = acc[2 * W], x1 = acc[2 * W + 1];
    if constexpr (RELU) {
        x0 = relu1(x0);
        x1 = relu1(x1);
    }
    out.v[W >> 2][W & 3] = __uint_as_float(cvt_pk_bf16(x0, x1));
}
template <int W0, int CNT, bool RELU>
__device__ __forceinline__ void convert_words_plain(const f32x16& acc, PTile& out) {
    static_for<CNT>([&](auto I) {
        constexpr int w = W0 + decltype(I)::value;
        if constexpr (w < 8) convert_word_plain<w, RELU>(acc, out);
    });
}
// PT point sets (32 points each) per wave share every weight fragment: the layer's MFMAs go
// (fragment, set 0), (fragment, set 1), ... so a ds_read_b128 feeds PT MFMAs.  At PT = 1 every
// 32-cycle MFMA needs its own 1 KiB fragment: 4 SIMDs x 32 B/cycle = the LDS's whole 128 B/cycle.
template <int NT, int STEPS, bool DEFER, int PT, class Out, class Def>
struct SidePlain {
    Out out;            // out(ic<q>) -> PTile* of set q
    Def deferred;       // deferred(ic<q>) -> PTile* (the previous layer's last tile of set q)
    f32x16* pend;       // pend[PT]
    f32x16 (*acc)[2];   // acc[PT][2]
    const float* bias_half;
    template <int T, int S, int H>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>, ic<H>) const {
        constexpr int C = (8 + STEPS - 1) / STEPS;
        constexpr int CD = (16 + STEPS - 1) / STEPS;  // deferred tile: done by STEPS/2, ahead of its first use
        static_for<PT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            if constexpr (H == 0) {
                if constexpr (T > 0) convert_words_plain<S * C, C, true>(pend[q], out(ic<q>{})[T - 1]);
                if constexpr (T == 0 && DEFER) convert_words_plain<S * CD, CD, true>(pend[q], *deferred(ic<q>{}));
            } else {
                // the next tile's bias: 4 quads spread over the pair-steps (2 per step when STEPS == 2)
                constexpr int QB = (4 + STEPS - 1) / STEPS;
                if constexpr (T + 1 < NT)
                    static_for<QB>([&](auto B_) {
                        constexpr int b = S * QB + decltype(B_)::value;
                        if constexpr (b < 4) bias_quad<b>(acc[q][(T + 1) & 1], bias_half + 32 * (T + 1));
                    });
            }
        });
    }
};

struct NoDeferred {
    template <int Q>
    __device__ __forceinline__ PTile* operator()(ic<Q>) const { return nullptr; }
};

template <int F0, int NT, int KS, bool DEFER, int PT, class Out, class Def, class BGet, class WS, class Hook = NoHook>
__device__ __forceinline__ void run_layer_plain(Out&& out, Def&& deferred, f32x16 (&pend)[PT], f32x16 (&acc)[PT][2],
                                                const float* bias_half, BGet&& bget, WS& ws, FragReader& fr,
                                                Hook&& after_open = NoHook{}) {
    constexpr int STEPS = KS / 2, NP = NT * STEPS;
    constexpr bool LAST = (F0 + NT * KS == kPlainUsedFrags);
    static_assert(F0 % 2 == 0 && KS % 2 == 0, "k-steps are consumed in pairs");
    static_for<PT>([&](auto Q_) { bias_tile(acc[decltype(Q_)::value][0], bias_half); });
    const SidePlain<NT, STEPS, DEFER, PT, std::decay_t<Out>, std::decay_t<Def>> side{out, deferred, &pend[0], &acc[0], bias_half};
    if constexpr (F0 % kSliceFrags == 0) {
        ws.open_slice();
        after_open();
        fr.pref0 = fr.template issue<F0>();
        fr.pref1 = fr.template issue<F0 + 1>();
    } else {
        static_assert(std::is_same_v<std::decay_t<Hook>, NoHook>, "a hook needs a layer that starts on a slice boundary");
    }
    f32x4 a0 = fr.pref0, a1 = fr.pref1;
    static_for<NP>([&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        constexpr int t = pi / STEPS, s = pi % STEPS;
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
        static_for<PT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            acc[q][t & 1] = mfma_bf(a0, bget(ic<q>{}, ic<2 * s>{}), acc[q][t & 1]);
        });
        side(ic<t>{}, ic<s>{}, ic<0>{});
        static_for<PT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            acc[q][t & 1] = mfma_bf(a1, bget(ic<q>{}, ic<2 * s + 1>{}), acc[q][t & 1]);
        });
        side(ic<t>{}, ic<s>{}, ic<1>{});
        if constexpr (s == STEPS - 1) static_for<PT>([&](auto Q_) { pend[decltype(Q_)::value] = acc[decltype(Q_)::value][t & 1]; });
        if constexpr (next_crosses && pi + 1 < NP) {
            ws.open_slice();
            n0 = fr.template issue<f + 2>();
            n1 = fr.template issue<f + 3>();
        }
        a0 = n0;
        a1 = n1;
    });
    fr.pref0 = a0;
    fr.pref1 = a1;
}

// NW waves per workgroup, PT point sets of 32 points per wave (32 NW PT points per pass).
//   NW = 8, PT = 1: two waves on each SIMD; the kernel fits 256 registers in this mode (a few spills), one
//     wave's conversions / encoding / piece issue run in the shadow of the other wave's MFMAs.  Same-box A/B
//     against NW = 4, PT = 1: +14 %.  The accumulators then live in VGPRs, so nothing that reads an MFMA
//     result may be inline asm (the hazard recogniser does not look inside).
//   NW = 4, PT = 2: one wave per SIMD with 512 registers; every weight fragment read from LDS feeds two
//     MFMAs (see SidePlain), and two independent accumulators alternate on the pipe.
template <int MODE, int NW, int PT>
__global__ __launch_bounds__(64 * NW, 1) void mlp_bf16_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_s = reinterpret_cast<float*>(smem + kRingFrags * kFragBytes);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    for (int i = tid; i < kBiasFloats; i += 64 * NW) bias_s[i] = a.bias[i];
    __syncthreads();

    Diag dg;
    WStreamT<NW> ws;
    ws.dg = &dg;
    ws.init(a.wstream, kPlainNumSlices, ring, tid, wave);
    PeLane pln;
    pln.init(h);
    FragReader fr;
    fr.addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    fr.addr1 = fr.addr0 + 64 * kFragBytes;
    const float* bias_h = bias_s + 4 * h;
    constexpr int kSets = NW * PT;            // 32-point sets per pass; wave w owns sets w PT .. w PT + PT - 1
    constexpr int kTilePts = 32 * kSets;
    const long ntiles = (a.n_points + kTilePts - 1) / kTilePts;

    PointIn cur[PT], nxt[PT];
    static_for<PT>([&](auto Q_) {
        constexpr int q = decltype(Q_)::value;
        load_point<MODE, kSets>(a, blockIdx.x, wave * PT + q, m, cur[q]);
        nxt[q] = cur[q];
    });
#ifdef IDN_TIMING_PE_ONCE   // timing-only (wrong results): the encoding of the block's first tile serves every pass
    f32x4 pe_v[PT][4], pd_v[PT][2];
#endif
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long P[PT];
        bool valid[PT];
#ifndef IDN_TIMING_PE_ONCE
        f32x4 pe_v[PT][4], pd_v[PT][2];
#endif
        auto pack_feats = [&](auto&& feat, f32x4* o, auto NKS) {
            static_for<decltype(NKS)::value>([&](auto S_) {
                constexpr int s = decltype(S_)::value;
                static_for<4>([&](auto W_) {
                    constexpr int w = decltype(W_)::value;
                    constexpr int j0 = 2 * w, j1 = 2 * w + 1;
                    constexpr int k0 = 16 * s + (j0 & 3) + 8 * (j0 >> 2), k1 = 16 * s + (j1 & 3) + 8 * (j1 >> 2);
                    o[s][w] = __uint_as_float(cvt_pk_bf16(feat(ic<k0>{}), feat(ic<k1>{})));
                });
            });
        };
        static_for<PT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            P[q] = tile * kTilePts + (wave * PT + q) * 32 + m;
            valid[q] = P[q] < a.n_points;
            const long Pc = valid[q] ? P[q] : a.n_points - 1;
#ifdef IDN_TIMING_PE_ONCE
            if (tile == blockIdx.x)
#endif
            input_features<MODE>(a, Pc, h, pln, cur[q], [&](auto&& fpt, auto&& fdir) {
                pack_feats(fpt, pe_v[q], ic<4>{});
                pack_feats(fdir, pd_v[q], ic<2>{});
            });
        });

        PTile A[PT][8], B[PT][8], V[PT][4], Vx[PT][5], none[PT][1];
        f32x16 acc[PT][2], pend[PT];
        // providers: set q's tile array / one tile of it / k-step s of its packed activations
        auto set_of = [](auto* arr) { return [arr](auto Q_) { return &arr[decltype(Q_)::value][0]; }; };
        auto last_of = [](auto* arr, auto T_) { return [arr](auto Q_) { return &arr[decltype(Q_)::value][decltype(T_)::value]; }; };
        auto tiles = [](auto* arr) {
            return [arr](auto Q_, auto S_) { constexpr int s = decltype(S_)::value; return arr[decltype(Q_)::value][s >> 1].v[s & 1]; };
        };
        auto pe_g = [&](auto Q_, auto S_) { return pe_v[decltype(Q_)::value][decltype(S_)::value]; };

        run_layer_plain<plain_f0(0), 8, 4, false, PT>(set_of(A), NoDeferred{}, pend, acc, bias_h + bias_off(0), pe_g, ws, fr);
#pragma unroll 1
        for (int l = 1; l <= 3; l += 2) {
            run_layer_plain<plain_f0(1), 8, 16, true, PT>(set_of(B), last_of(A, ic<7>{}), pend, acc, bias_h + l * 256, tiles(A), ws, fr);
            run_layer_plain<plain_f0(2), 8, 16, true, PT>(set_of(A), last_of(B, ic<7>{}), pend, acc, bias_h + (l + 1) * 256, tiles(B), ws, fr);
        }
        run_layer_plain<plain_f0(5), 8, 20, true, PT>(
            set_of(B), last_of(A, ic<7>{}), pend, acc, bias_h + bias_off(5),
            [&](auto Q_, auto S_) {
                constexpr int q = decltype(Q_)::value, s = decltype(S_)::value;
                if constexpr (s < 4) return pe_v[q][s]; else return A[q][(s - 4) >> 1].v[(s - 4) & 1];
            },
            ws, fr);
        run_layer_plain<plain_f0(6), 8, 16, true, PT>(set_of(A), last_of(B, ic<7>{}), pend, acc, bias_h + bias_off(6), tiles(B), ws, fr, [&]() {
            static_for<PT>([&](auto Q_) {
                constexpr int q = decltype(Q_)::value;
                load_point<MODE, kSets>(a, tile + gridDim.x, wave * PT + q, m, nxt[q]);
            });
        });
        run_layer_plain<plain_f0(7), 8, 16, true, PT>(set_of(B), last_of(A, ic<7>{}), pend, acc, bias_h + bias_off(7), tiles(A), ws, fr,
                                                      [&]() { static_for<PT>([&](auto Q_) { touch_point(nxt[decltype(Q_)::value]); }); });
        run_layer_plain<plain_f0(8), 5, 18, true, PT>(
            set_of(Vx), last_of(B, ic<7>{}), pend, acc, bias_h + bias_off(8),
            [&](auto Q_, auto S_) {
                constexpr int q = decltype(Q_)::value, s = decltype(S_)::value;
                if constexpr (s < 16) return B[q][s >> 1].v[s & 1]; else return pd_v[q][s - 16];
            },
            ws, fr);
        float sigma[PT];
        static_for<PT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            sigma[q] = pend[q][0];
            static_for<4>([&](auto T) { V[q][decltype(T)::value] = Vx[q][decltype(T)::value]; });
        });
        run_layer_plain<plain_f0(9), 4, 8, false, PT>(set_of(A), NoDeferred{}, pend, acc, bias_h + bias_off(9), tiles(V), ws, fr);
        run_layer_plain<plain_f0(10), 4, 8, true, PT>(set_of(V), last_of(A, ic<3>{}), pend, acc, bias_h + bias_off(10), tiles(A), ws, fr);
        run_layer_plain<plain_f0(11), 1, 8, true, PT>(set_of(none), last_of(V, ic<3>{}), pend, acc, bias_h + bias_off(11), tiles(V), ws, fr);
        finish_pass<kPlainUsedFrags, kPlainStreamFrags>(ws);

        static_for<PT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            if (valid[q] && h == 0) {
                f32x4 o;
                o.x = pend[q][0];
                o.y = pend[q][1];
                o.z = pend[q][2];
                o.w = sigma[q];
                *reinterpret_cast<f32x4*>(a.raw + P[q] * 4) = o;
            }
            cur[q] = nxt[q];
        });
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}

#ifndef IDN_PLAIN_WAVES
#define IDN_PLAIN_WAVES 8
#endif
#ifndef IDN_PLAIN_SETS
#define IDN_PLAIN_SETS 1
#endif
int launch_mlp_bf16(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                    const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s) {
    constexpr int NW = IDN_PLAIN_WAVES, PT = IDN_PLAIN_SETS;
    if (n_points <= 0) return IDN_OK;
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bf16_kernel<kModeRays, NW, PT>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bf16_kernel<kModeX, NW, PT>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bf16_kernel<kModePts, NW, PT>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ntiles = (n_points + 32 * NW * PT - 1) / (32 * NW * PT);
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    MlpArgs a{packed, folded, x, rays, z, pts, dirs, (long)n_points, n_samples, raw, nullptr, 0};
    ProfScope prof(s, n_points);
    if (x)
        hipLaunchKernelGGL((mlp_bf16_kernel<kModeX, NW, PT>), dim3(grid), dim3(64 * NW), kMlpLds, s, a);
    else if (pts)
        hipLaunchKernelGGL((mlp_bf16_kernel<kModePts, NW, PT>), dim3(grid), dim3(64 * NW), kMlpLds, s, a);
    else
        hipLaunchKernelGGL((mlp_bf16_kernel<kModeRays, NW, PT>), dim3(grid), dim3(64 * NW), kMlpLds, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

#endif  // IDN_X3_FP16

int X3_LAUNCH(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                      const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s) {
    if (n_points <= 0) return IDN_OK;
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&X3_KERNEL<kModeRays>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&X3_KERNEL<kModeX>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&X3_KERNEL<kModePts>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLds));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ntiles = (n_points + 127) / 128;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    MlpArgs a{packed, folded, x, rays, z, pts, dirs, (long)n_points, n_samples, raw, nullptr, 0};
    ProfScope prof(s, n_points);
    if (x)
        hipLaunchKernelGGL((X3_KERNEL<kModeX>), dim3(grid), dim3(256), kMlpLds, s, a);
    else if (pts)
        hipLaunchKernelGGL((X3_KERNEL<kModePts>), dim3(grid), dim3(256), kMlpLds, s, a);
    else
        hipLaunchKernelGGL((X3_KERNEL<kModeRays>), dim3(grid), dim3(256), kMlpLds, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
