// Fused positional-encoding + FaceNeRF MLP forward on the bf16 matrix pipe at fp32-grade accuracy (gfx950):
// every fp32 operand -- weight or activation -- is the exact sum of three bf16 pieces
//      x = p1 + p2 + p3,   p1 = bf16(x), p2 = bf16(x - p1), p3 = bf16(x - p1 - p2)      (round to nearest),
// 3 x 8 significand bits in fp32's exponent range (nothing is scaled, nothing fp32 holds overflows), and a
// product keeps the six piece products down to 2^-16 of it,
//      a.b ~ a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1        (dropped: <= 2^-23 |a b|),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: the rounding of an fp32 fma chain at 16 / 6 of the fp32 MFMA
// rate.  (bf16x3 keeps two pieces per operand and three products: 1e-5; the third piece is what makes this one
// fp32-grade.  The training step's 256 x 256 weight-gradient GEMMs use the same arithmetic, train.hip.)
//
// K-MAJOR layers (round 3).  A layer walks its 16-channel k-steps in order and, inside a k-step, its n-tiles: step
// (s, t) = six piece products of k-step s into accumulator tile t.  The B operand of a k-step -- the three pieces of 16
// input channels of the wave's 32 points -- is therefore live for ONE k-step only, and it is produced where it is
// needed: while k-step s runs on the matrix pipe, the vector unit ReLUs and splits the eight accumulator registers of
// the previous layer that k-step s + 1 consumes (one word pair at a time, in the MFMA shadow).  Two sets of eight
// accumulator tiles alternate as output and input of consecutive layers; the activations never exist as 192 registers
// of pieces, and no conversion sits between layers.  (Round 2 was tile-major: a layer's whole input as pieces, its
// output as accumulators, and 13 % of the pass in the conversions at the layer ends.)
//
// Weight stream: per step a TRIPLE of fragments (p1, p2, p3) in 48-fragment ring slots: a slice is 16 steps, every
// layer starts where it does in the other streams modulo the ring, and the 3.375 MiB stream stays in the per-XCD L2.
#include "mlp_x6.h"

namespace idn {
namespace x6 {

constexpr int f0(int l) { return kX6KFrags * plain_f0(l); }   // triples: 3 fragments per step (plain_f0 counts steps)

// The training variant records a half tile where it is converted: the "unit off" bits of the eight pre-activations into the
// layer's mask words, the ReLU applied in place, two 16-byte stores into this lane's row of the layer's activation
// matrix (register 4 q + j of tile T is channel 32 T + 8 q + 4 h + j), then the split.
struct Recorder {
    uint32_t* mk;     // the mask words of the layer being recorded (dword T / 2: tiles T, T + 1, value r at bit 31 - (16 (T & 1) + r))
    float* row;       // first float of this lane's row of that layer's matrix + 4 h
};

template <int T, int HS, int W, bool SAVE>
__device__ __forceinline__ void convert_word(f32x16& acc, KP& out, const Recorder& rec) {
    constexpr int r0 = 8 * HS + 2 * W;
    if constexpr (SAVE) {
        acc[r0] = relu1(acc[r0]);
        acc[r0 + 1] = relu1(acc[r0 + 1]);
        // "unit off" = pre-activation <= 0 (+0.0 included, as torch's relu backward): the sign bit of the ReLU'd pattern - 1
        rec.mk[T >> 1] = __builtin_amdgcn_alignbit(rec.mk[T >> 1], __float_as_uint(acc[r0]) - 1u, 31);
        rec.mk[T >> 1] = __builtin_amdgcn_alignbit(rec.mk[T >> 1], __float_as_uint(acc[r0 + 1]) - 1u, 31);
        float w1, w2, w3;
        split3(acc[r0], acc[r0 + 1], w1, w2, w3);
        out.p[0][W] = w1;
        out.p[1][W] = w2;
        out.p[2][W] = w3;
    } else {
        float w1, w2, w3;
        split3(relu1(acc[r0]), relu1(acc[r0 + 1]), w1, w2, w3);
        out.p[0][W] = w1;
        out.p[1][W] = w2;
        out.p[2][W] = w3;
    }
}
template <int T, int Q>
__device__ __forceinline__ void store_quad(const f32x16& acc, const Recorder& rec) {
#ifdef IDN_TIMING_NO_ROW_STORES   // timing-only experiment (wrong results): what do the row stores cost?
    return;
#endif
    *reinterpret_cast<f32x4*>(rec.row + 32 * T + 8 * Q) = f32x4{acc[4 * Q], acc[4 * Q + 1], acc[4 * Q + 2], acc[4 * Q + 3]};
}

// The side work of step (.., t) of a k-step of NT tiles that prepares half tile (T, HS) of `acc` as the next k-step's
// pieces: words at conv_slot(NT, 0..3); training: the two row stores of the half tile at store_slot(NT, 0 / 1).
template <int NT, int t, int T, int HS, bool SAVE>
__device__ __forceinline__ void prepare_half(f32x16& acc, KP& out, const Recorder& rec) {
    static_for<4>([&](auto W_) {
        constexpr int w = decltype(W_)::value;
        if constexpr (conv_slot(NT, w) == t) convert_word<T, HS, w, SAVE>(acc, out, rec);
    });
    if constexpr (SAVE) {
        static_assert(store_slot(NT, 0) >= conv_slot(NT, 1) && store_slot(NT, 1) >= conv_slot(NT, 3), "a quad is stored after its words");
        if constexpr (t == store_slot(NT, 0)) store_quad<T, 2 * HS>(acc, rec);
        if constexpr (t == store_slot(NT, 1)) store_quad<T, 2 * HS + 1>(acc, rec);
    }
}
// a hidden layer (kernel body, `hidden`): k-steps S0 .. KS - 1 come from the input accumulators (prepared during k-steps
// S0 - 1 .. KS - 2), and its last k-step prepares the next layer's first unless that one is an encoding k-step
template <int NT, int KS, int S0, bool NEXT_PE, bool SAVE>
using HiddenStoresT = Stores<NT, (S0 > 0 ? S0 : 1), (NEXT_PE ? KS : KS + 1), SAVE>;
// views_linears.0: k-steps 1..15 from the trunk, 16 and 17 direction pieces, and its last k-step prepares views_linears.1's first
template <class ST_, int NP_>
struct PrevLayer {   // the layer in front of a layer: its row stores and its number of steps (run_layer's PST, PNP)
    using ST = ST_;
    static constexpr int NP = NP_;
};
template <bool SAVE>
struct Views0StoresT {
    static constexpr int at(int s, int t) { return Stores<5, 1, 16, SAVE>::at(s, t) + Stores<5, 18, 19, SAVE>::at(s, t); }
    static constexpr int in_steps(int i0, int i1, int np) {
        int n = 0;
        for (int i = (i0 < 0 ? 0 : i0); i <= i1 && i < np; ++i) n += at(i / 5, i % 5);
        return n;
    }
};

template <int MODE, bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_bf16x6_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    using WS = WStream6x3;   // three ring slots: pieces fetched two slices ahead
    float* bias_s = reinterpret_cast<float*>(smem + WS::kSlotsT * kX6SliceFrags * kFragBytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;

    for (int i = tid; i < kBiasFloats; i += 256) bias_s[i] = a.bias[i];
    __syncthreads();  // the bias block is read (by other waves) before the first slice barrier

    Diag dg;
    WS ws;
    ws.dg = &dg;
#ifdef IDN_TIMING_STREAM_WRAP   // timing-only (wrong results): the stream wraps after this many slices -- is anything left of the L2 question?
    ws.init(a.wstream, IDN_TIMING_STREAM_WRAP, ring, tid, wave);
#else
    ws.init(a.wstream, kX6NumSlices, ring, tid, wave);
#endif
    PeLane pln;
    pln.init(h);

    FragReader fr;
    fr.addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    fr.addr1 = fr.addr0 + 64 * kFragBytes;
    fr.addr2 = fr.addr0 + 128 * kFragBytes;
    const float* bias_h = bias_s + 4 * h;
    const long ntiles = (a.n_points + 127) >> 7;

    PointIn cur, nxt;
    load_point<MODE>(a, blockIdx.x, wave, m, cur);
    nxt = cur;
    f32x4 pref[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        DIAG_ONLY(const unsigned long long t_tile = clock64(); dg.begin();)
        const long P = tile * 128 + wave * 32 + m;
        const bool valid = P < a.n_points;
        const long Pc = valid ? P : a.n_points - 1;

        // ---- inputs: this lane's half of the 64 point features and 32 direction features, as pieces.
        // Element j of k-step s, lane half h = feature 16 s + (j & 3) + 8 (j >> 2) + 4 h.
        f32x4 pe_p[3][4], pd_p[3][2];
        {
            // feature 8 g + 4 h + j of the point (g < 8) and of the direction (g < 4)
            float pe_f[8][4], pd_f[4][4];
            input_features<MODE>(a, Pc, h, pln, cur, [&](auto&& fpt, auto&& fdir) {
                static_for<8>([&](auto G) {
                    static_for<4>([&](auto J) {
                        constexpr int g = decltype(G)::value, j = decltype(J)::value;
                        pe_f[g][j] = fpt(ic<8 * g + j>{});
                        if constexpr (g < 4) pd_f[g][j] = fdir(ic<8 * g + j>{});
                    });
                });
            });
            if constexpr (SAVE) {   // every row of the slab, padding rows included (they repeat the last point)
                float* x0 = a.acts + act_off(kActX0) * a.p_pad + P * 64 + 4 * h;
                float* dr = a.acts + act_off(kActDir) * a.p_pad + P * 64 + 4 * h;
                static_for<8>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    *reinterpret_cast<f32x4*>(x0 + 8 * g) = f32x4{pe_f[g][0], pe_f[g][1], pe_f[g][2], pe_f[g][3]};
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (g < 4) v = f32x4{pd_f[g][0], pd_f[g][1], pd_f[g][2], pd_f[g][3]};
                    *reinterpret_cast<f32x4*>(dr + 8 * g) = v;
                });
            }
            // word w of k-step s = elements (2 w, 2 w + 1) = features 16 s + 8 (w >> 1) + 2 (w & 1) + {0, 1} (+ 4 h)
            auto pack_feats = [&](auto& f, auto& out, auto NKS) {
                static_for<decltype(NKS)::value>([&](auto S_) {
                    constexpr int s = decltype(S_)::value;
                    static_for<4>([&](auto W_) {
                        constexpr int w = decltype(W_)::value;
                        float w1, w2, w3;
                        split3(f[2 * s + (w >> 1)][2 * (w & 1)], f[2 * s + (w >> 1)][2 * (w & 1) + 1], w1, w2, w3);
                        out[0][s][w] = w1;
                        out[1][s][w] = w2;
                        out[2][s][w] = w3;
                    });
                });
            };
            pack_feats(pe_f, pe_p, ic<4>{});
            pack_feats(pd_f, pd_p, ic<2>{});
        }
        DIAG_END(dg, kDgInput);

        f32x16 X[8], Y[8];   // two sets of accumulator tiles: output and input of consecutive layers, alternating
        using Views0Stores = Views0StoresT<SAVE>;
        KP B;
        uint32_t mk[4] = {0u, 0u, 0u, 0u};   // SAVE: the ReLU mask bits of the layer being recorded
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        auto pe_kstep = [&](auto S_, KP& dst) {
            constexpr int s = decltype(S_)::value;
            dst.p[0] = pe_p[0][s];
            dst.p[1] = pe_p[1][s];
            dst.p[2] = pe_p[2][s];
        };
        auto pd_kstep = [&](auto S_, KP& dst) {
            constexpr int s = decltype(S_)::value;
            dst.p[0] = pd_p[0][s];
            dst.p[1] = pd_p[1][s];
            dst.p[2] = pd_p[2][s];
        };
        auto act_row = [&](int idx, int ld) { return SAVE ? a.acts + (long)act_off(idx) * a.p_pad + P * ld + 4 * h : nullptr; };
        auto mask_store = [&](int idx) {   // the finished mask words of layer `idx`; the next layer's start from zero
            if constexpr (SAVE) {
                u32x4* mp = reinterpret_cast<u32x4*>(a.acts + (size_t)kActCols * a.p_pad) + mask_index(idx - kActA1, a.p_pad, tile * 4 + wave, lane);
                *mp = u32x4{mk[0], mk[1], mk[2], mk[3]};
                mk[0] = mk[1] = mk[2] = mk[3] = 0u;
            }
        };
        // the next layer's biases, loaded into the set its input no longer occupies, one tile per step of this layer's last k-step
        auto bias_next = [&](auto T_, auto NTnext, f32x16* On, const float* bias_l) {
            constexpr int t = decltype(T_)::value;
            if constexpr (t < decltype(NTnext)::value) bias_tile(On[t], bias_l + 32 * t);
        };

        // ---------------------------------------------------------------------------------------------------------
        // A hidden layer whose input is the previous layer's accumulators `In` (from k-step S0 on; k-steps [0, S0) are
        // point-encoding pieces), output tiles `Out`.  During k-step s the vector unit prepares k-step s + 1:
        //   s + 1 <  S0 : the encoding pieces (a copy);
        //   s + 1 >= S0 : half tile ((s + 1 - S0) / 2, (s + 1 - S0) % 2) of In -- recorded (training) under activation
        //                 index `in_idx` with rows of `in_ld` floats;
        //   s + 1 == KS : the NEXT layer's first k-step -- half tile (0, 0) of Out (recorded under out_idx), unless that
        //                 layer starts with encoding pieces (next_pe) -- and that layer's biases into `In`.
        // ---------------------------------------------------------------------------------------------------------
        auto hidden = [&](auto F0c, auto NTc, auto KSc, auto S0c, auto Prev, f32x16* In, f32x16* Out, int in_idx, int in_ld, int out_idx, int out_ld,
                          auto NTnext, const float* bias_nxt, auto next_pe, auto&& tail_pieces, auto&& hook) __attribute__((always_inline)) {
            constexpr int F0 = decltype(F0c)::value, NT = decltype(NTc)::value, KS = decltype(KSc)::value, S0 = decltype(S0c)::value;
            constexpr bool NEXT_PE = decltype(next_pe)::value;
            const Recorder rin{mk, act_row(in_idx, in_ld)}, rout{mk, act_row(out_idx, out_ld)};
            auto side = [&](auto S_, auto T_, KP& Bn) {
                constexpr int s = decltype(S_)::value, t = decltype(T_)::value, sn = s + 1;
                if constexpr (sn < S0) {
                    if constexpr (t == 0) pe_kstep(ic<sn>{}, Bn);
                } else if constexpr (sn < KS) {
                    constexpr int T = (sn - S0) >> 1, HS = (sn - S0) & 1;
                    prepare_half<NT, t, T, HS, SAVE>(In[T], Bn, rin);
                    // the mask words of the layer being recorded are complete with its last half tile
                    if constexpr (SAVE && sn == KS - 1 && t == NT - 1) mask_store(in_idx);
                } else {
                    bias_next(T_, NTnext, In, bias_nxt);
                    if constexpr (NEXT_PE) {
                        if constexpr (t == 0) tail_pieces(Bn);
                    } else {
                        prepare_half<NT, t, 0, 0, SAVE>(Out[0], Bn, rout);
                    }
                }
            };
            using P = decltype(Prev);
            run_layer<F0, NT, KS, (F0 + kX6KFrags * NT * KS == kX6UsedFrags), typename P::ST, P::NP, HiddenStoresT<NT, KS, S0, NEXT_PE, SAVE>>(Out, B, side, ws, fr, pref, hook);
        };

        // ---- biases of pts_linears.0, pieces of its first k-step
        static_for<8>([&](auto T_) { bias_tile(X[decltype(T_)::value], bias_h + bias_off(0) + 32 * decltype(T_)::value); });
        pe_kstep(ic<0>{}, B);
        settle(B);   // (the other encoding pieces are first read k-steps later)
        const auto no_tail = [](KP&) {};
        // the layer before each layer (its row stores and its number of steps): what may still be in flight at a slice barrier
        using PrevNone = PrevLayer<NoStores, 1>;                                           // (the pass before: nothing counted)
        using PrevL0 = PrevLayer<HiddenStoresT<8, 4, 4, false, SAVE>, 8 * 4>;              // pts_linears.0: only its last k-step prepares
        using PrevH = PrevLayer<HiddenStoresT<8, 16, 0, false, SAVE>, 8 * 16>;             // a 256 x 256 layer
        using PrevL4 = PrevLayer<HiddenStoresT<8, 16, 0, true, SAVE>, 8 * 16>;             // pts_linears.4: its last k-step hands over encoding pieces
        using PrevL5 = PrevLayer<HiddenStoresT<8, 20, 4, false, SAVE>, 8 * 20>;
        using PrevV0 = PrevLayer<Views0Stores, 5 * 18>;
        using PrevV = PrevLayer<HiddenStoresT<4, 8, 0, false, SAVE>, 4 * 8>;
        // ---- pts_linears.0 : PE(64) -> 256 (into X).  Its k-steps are encoding pieces; its last one prepares (X[0], half 0).
        hidden(ic<f0(0)>{}, ic<8>{}, ic<4>{}, ic<4>{}, PrevNone{}, Y, X, kActA1, 256, kActA1 + 0, 256, ic<8>{}, bias_h + bias_off(1), ic<0>{}, no_tail, NoHook{});
        // ---- pts_linears.1..4
        hidden(ic<f0(1)>{}, ic<8>{}, ic<16>{}, ic<0>{}, PrevL0{}, X, Y, kActA1 + 0, 256, kActA1 + 1, 256, ic<8>{}, bias_h + bias_off(2), ic<0>{}, no_tail, NoHook{});
        hidden(ic<f0(2)>{}, ic<8>{}, ic<16>{}, ic<0>{}, PrevH{}, Y, X, kActA1 + 1, 256, kActA1 + 2, 256, ic<8>{}, bias_h + bias_off(3), ic<0>{}, no_tail, NoHook{});
        hidden(ic<f0(3)>{}, ic<8>{}, ic<16>{}, ic<0>{}, PrevH{}, X, Y, kActA1 + 2, 256, kActA1 + 3, 256, ic<8>{}, bias_h + bias_off(4), ic<0>{}, no_tail, NoHook{});
        // pts_linears.4's output (a5) is not the first thing pts_linears.5 reads: its first four k-steps are the encoding
        hidden(ic<f0(4)>{}, ic<8>{}, ic<16>{}, ic<0>{}, PrevH{}, Y, X, kActA1 + 3, 256, kActA1 + 4, 256, ic<8>{}, bias_h + bias_off(5), ic<1>{},
               [&](KP& Bn) { pe_kstep(ic<0>{}, Bn); }, NoHook{});
        // ---- pts_linears.5 : [PE(64) | 256] -> 256 (X -> Y)
        hidden(ic<f0(5)>{}, ic<8>{}, ic<20>{}, ic<4>{}, PrevL4{}, X, Y, kActA1 + 4, 256, kActA1 + 5, 256, ic<8>{}, bias_h + bias_off(6), ic<0>{}, no_tail,
               [&]() { load_point<MODE>(a, tile + gridDim.x, wave, m, nxt); });
        // ---- pts_linears.6, .7
        hidden(ic<f0(6)>{}, ic<8>{}, ic<16>{}, ic<0>{}, PrevL5{}, Y, X, kActA1 + 5, 256, kActA1 + 6, 256, ic<8>{}, bias_h + bias_off(7), ic<0>{}, no_tail, [&]() { touch_point(nxt); });
        hidden(ic<f0(7)>{}, ic<8>{}, ic<16>{}, ic<0>{}, PrevH{}, X, Y, kActA1 + 6, 256, kActA1 + 7, 256, ic<5>{}, bias_h + bias_off(8), ic<0>{}, no_tail, NoHook{});
        // ---- views_linears.0 (+ alpha_linear as channel 128): [256 | dirPE(32)] -> 160 (Y -> X[0..4]).
        //      Tiles 0..3 are hidden units; tile 4 is not: its row 0 is sigma.  k-steps 16, 17 are direction pieces.
        {
            constexpr int F0 = f0(8), NT = 5, KS = 18;
            const Recorder rin{mk, act_row(kActA1 + 7, 256)}, rout{mk, act_row(kActV1, 128)};
            auto side = [&](auto S_, auto T_, KP& Bn) {
                constexpr int s = decltype(S_)::value, t = decltype(T_)::value, sn = s + 1;
                if constexpr (sn < 16) {
                    prepare_half<NT, t, (sn >> 1), (sn & 1), SAVE>(Y[sn >> 1], Bn, rin);
                    if constexpr (SAVE && sn == 15 && t == NT - 1) mask_store(kActA1 + 7);
                } else if constexpr (sn < KS) {
                    if constexpr (t == 0) pd_kstep(ic<sn - 16>{}, Bn);
                } else {
                    bias_next(T_, ic<4>{}, Y, bias_h + bias_off(9));
                    prepare_half<NT, t, 0, 0, SAVE>(X[0], Bn, rout);
                }
            };
            run_layer<F0, NT, KS, false, typename PrevH::ST, PrevH::NP, Views0Stores>(X, B, side, ws, fr, pref, NoHook{});
        }
        const float sigma = X[4][0];   // channel 128 = tile 4, register 0, lane half 0
        // ---- views_linears.1, .2 : 128 -> 128
        hidden(ic<f0(9)>{}, ic<4>{}, ic<8>{}, ic<0>{}, PrevV0{}, X, Y, kActV1, 128, kActV1 + 1, 128, ic<4>{}, bias_h + bias_off(10), ic<0>{}, no_tail, NoHook{});
        hidden(ic<f0(10)>{}, ic<4>{}, ic<8>{}, ic<0>{}, PrevV{}, Y, X, kActV1 + 1, 128, kActV1 + 2, 128, ic<1>{}, bias_h + bias_off(11), ic<0>{}, no_tail, NoHook{});
        // ---- rgb_linear : 128 -> 3 (rows 0..2 of one tile, Y[0]); its k-steps record v3
        {
            const Recorder rin{mk, act_row(kActV1 + 2, 128)};
            auto side = [&](auto S_, auto T_, KP& Bn) {
                constexpr int sn = decltype(S_)::value + 1;
                if constexpr (sn < 8) {
                    prepare_half<1, 0, (sn >> 1), (sn & 1), SAVE>(X[sn >> 1], Bn, rin);
                    if constexpr (SAVE && sn == 7) mask_store(kActV1 + 2);
                }
            };
            run_layer<f0(11), 1, 8, true, typename PrevV::ST, PrevV::NP, Stores<1, 1, 8, SAVE>>(Y, B, side, ws, fr, pref, NoHook{});
        }
        finish_pass6<kX6UsedFrags, kX6StreamFrags>(ws);

        if (valid && h == 0) {
            f32x4 o;
            o.x = Y[0][0];
            o.y = Y[0][1];
            o.z = Y[0][2];
            o.w = sigma;
            *reinterpret_cast<f32x4*>(a.raw + P * 4) = o;
        }
        DIAG_ONLY(dg.acc[kDgTotal] += clock64() - t_tile;)
        cur = nxt;
    }
#ifdef IDN_DIAG   // diagnostic build only: per-wave cycle totals by category (tools/diag_mlp_x6.py)
    if (lane == 0)
        for (int c = 0; c < 5; ++c) atomicAdd(&g_diag[c], dg.acc[c]);
    if (lane == 0) atomicAdd(&g_diag[5], 1ull);
#endif
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}

}  // namespace x6

#ifdef IDN_DIAG
extern "C" int idealnerf_diag_read_x6(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_diag), 8 * sizeof(unsigned long long)) != hipSuccess) return -3;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof(z)) != hipSuccess) return -3;
    return 0;
}
#endif

int launch_mlp_bf16x6(const float* packed, const float* folded, const float* x, const float* rays, const float* z,
                      const float* pts, const float* dirs, int64_t n_points, int n_samples, float* raw, hipStream_t s,
                      float* acts, int64_t p_pad) {
    if (n_points <= 0) return IDN_OK;
    if (acts && (x || pts)) return fail(IDN_EUNSUPPORTED, "the activation-saving forward takes rays");
    static LaunchSetup setup;
    int num_cu = 0;
    constexpr int kLdsInfer = x6::mlp_lds6<x6::WStream6x3>(), kLdsTrain = kLdsInfer;
    static_assert(kLdsInfer <= 160 * 1024, "LDS per CU");
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModeRays, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsInfer));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModeRays, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTrain));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModeX, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsInfer));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModePts, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsInfer));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ntiles = (n_points + 127) / 128;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    MlpArgs a{packed, folded, x, rays, z, pts, dirs, (long)n_points, n_samples, raw, acts, (long)p_pad};
    ProfScope prof(s, n_points, acts ? IDN_PROF_MLP_FWD_SAVE_X6 : IDN_PROF_MLP_FWD);
    if (x)
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModeX, false>), dim3(grid), dim3(256), kLdsInfer, s, a);
    else if (pts)
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModePts, false>), dim3(grid), dim3(256), kLdsInfer, s, a);
    else if (acts)
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModeRays, true>), dim3(grid), dim3(256), kLdsTrain, s, a);
    else
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModeRays, false>), dim3(grid), dim3(256), kLdsInfer, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
