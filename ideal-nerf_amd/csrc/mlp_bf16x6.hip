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
// Same computation, stream machinery and tile-major layer order as the other MLP kernels.  What differs is where a
// layer's data lives: the INPUT of a layer is held as pieces (8 tiles x 3 pieces x 2 k-steps x 4 registers = 192
// registers per wave of 32 points), its OUTPUT as the fp32 accumulator tiles themselves (8 x 16 = 128 registers,
// like the fp32 kernel); at the end of a layer the accumulators are ReLU'd and split into the piece registers,
// which the finished layer no longer needs.  (Input and output both as pieces -- the bf16x3 kernel's two ping-pong
// sets -- would be 384 registers before accumulators and encodings.)
//
// Weight stream: per (n-tile, 16-channel k-step) a TRIPLE of fragments (p1, p2, p3) in 48-fragment ring slots: a slice is
// 16 k-steps, as in the bf16x3 stream (16 pairs), so every layer starts where it does there modulo the ring, slices hold
// whole triples, and the 3.375 MiB stream stays in the per-XCD L2 (idn_internal.h).
#include "mlp_x6.h"

namespace idn {
namespace x6 {

constexpr int f0(int l) { return kX6KFrags * plain_f0(l); }   // triples: 3 fragments per k-step (plain_f0 counts k-steps)

// One layer, tile-major: for each n-tile t, KS k-steps of six piece products into O[t].  bget(ic<q>, ic<s>) =
// piece q of k-step s of the layer's input.  The first fragments arrive in `pref` (issued by the layer before, or
// here when the layer starts on a slice boundary) and leave in it for the next layer, VALID (retired): the
// conversion between two layers is long, and a register with a read in flight must not be moved or spilled.
struct NoTileSide {
    template <int T, int S>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>) const {}
};
template <int F0, int NT, int KS, int OPEN_YOUNGER = 0, int MID_YOUNGER = 0, class BGet, class Side = NoTileSide, class Hook = NoHook>
__device__ __forceinline__ void run_layer(f32x16* O, const float* bias_half, BGet&& bget, WStream6& ws, FragReader& fr,
                                          f32x4 (&pref)[3], Side&& side = NoTileSide{}, Hook&& after_open = NoHook{}) {
    constexpr int NP = NT * KS;
    constexpr bool LAST = (F0 + kX6KFrags * NP == kX6UsedFrags);
    static_assert(F0 % kX6KFrags == 0 && KS >= 4, "triples; the next tile's bias rides on the first four k-steps");
    bias_tile(O[0], bias_half);
    if constexpr (F0 % kX6SliceFrags == 0) {
        ws.template open_slice<OPEN_YOUNGER>();   // (training: the row stores of the layer before stay in flight)
        after_open();
        static_for<3>([&](auto Q) { pref[decltype(Q)::value] = issue6<F0 + decltype(Q)::value>(fr); });
        retire3<0>(pref);
    } else {
        static_assert(std::is_same_v<std::decay_t<Hook>, NoHook>, "a hook needs a layer that starts on a slice boundary");
    }
    f32x4 a[3] = {pref[0], pref[1], pref[2]};
    static_for<NP>([&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        constexpr int t = pi / KS, s = pi % KS;
        constexpr int f = F0 + kX6KFrags * pi;
        constexpr bool next_crosses = ((f + kX6KFrags) % kX6SliceFrags == 0);
        constexpr bool has_next = !(LAST && pi + 1 == NP);
        f32x4 n[3] = {a[0], a[1], a[2]};
        if constexpr (!next_crosses && has_next) {
            static_for<3>([&](auto Q) { n[decltype(Q)::value] = issue6<f + kX6KFrags + decltype(Q)::value>(fr); });
            if constexpr (pi > 0) retire3<3>(a);   // (step 0's arrived retired)
        } else {
            if constexpr (pi > 0) retire3<0>(a);
        }
        step_pieces6<f>(ws);
        const f32x4 b1 = bget(ic<0>{}, ic<s>{}), b2 = bget(ic<1>{}, ic<s>{}), b3 = bget(ic<2>{}, ic<s>{});
        O[t] = mfma_bf(a[0], b1, O[t]);   // w1 a1
        O[t] = mfma_bf(a[0], b2, O[t]);   // w1 a2
        O[t] = mfma_bf(a[1], b1, O[t]);   // w2 a1
        if constexpr (t + 1 < NT && s < 4) bias_quad<s>(O[t + 1], bias_half + 32 * (t + 1));   // the next tile starts from its bias
        O[t] = mfma_bf(a[1], b2, O[t]);   // w2 a2
        side(ic<t>{}, ic<s>{});           // training, 256 x 256 layers: tile t - 1 is recorded in this tile's second half (RecordInShadow)
        O[t] = mfma_bf(a[0], b3, O[t]);   // w1 a3
        O[t] = mfma_bf(a[2], b1, O[t]);   // w3 a1
        if constexpr (next_crosses && pi + 1 < NP) {
            // MID_YOUNGER (training, layers whose tiles are slices): the row stores issued in the second half of the slice
            // that ends here -- none in tile 0 -- are younger than the pieces of the slice being opened
            ws.template open_slice<(t >= 1 ? MID_YOUNGER : 0)>();
            static_for<3>([&](auto Q) { n[decltype(Q)::value] = issue6<f + kX6KFrags + decltype(Q)::value>(fr); });
        }
        a[0] = n[0];
        a[1] = n[1];
        a[2] = n[2];
    });
    // hand over retired fragments (those of the next layer's first k-step, when this layer ends inside a slice)
    if constexpr (!LAST && (F0 + kX6KFrags * NP) % kX6SliceFrags != 0) retire3<0>(a);
    pref[0] = a[0];
    pref[1] = a[1];
    pref[2] = a[2];
}

// Training: a tile is RECORDED once its accumulator is complete -- the sign bits of the 16 pre-activations into the
// layer's mask word, the ReLU applied in place, the 16 values into this lane's row of the layer's activation matrix
// (`row` = its first float + 4 h; register 4 q + j of tile t is channel 32 t + 8 q + 4 h + j, so a quad of registers is
// 16 contiguous bytes of the row).  On gfx9 stores count in vmcnt, vmcnt retires in issue order, and every slice barrier
// waits for the wave's pieces of the next slice: a store issued BEFORE those pieces delays the barrier until it has
// reached L2.  RecordSide is the plain form, for layers whose tiles are not slices: all tiles after the layer, and all
// of them before any is converted, so that the burst drains behind the conversion's ~3 000 cycles of vector work.
// (Tile t - 1 at steps 0..2 of tile t -- older than the pieces -- made every barrier of the layer wait: 4.96 -> 5.36 ms.)
struct RecordSide {
    f32x16* O;
    uint32_t* mk;
    float* row;
    float* lin;   // timing-only experiment: first float of this wave's 32 rows + 4 lane
    template <int T>
    __device__ __forceinline__ void signs_relu(ic<T>) const {
        collect_signs<T>(O[T], mk);
        relu_regs<0, 16>(O[T]);
    }
    template <int T, int Q0>
    __device__ __forceinline__ void store2(ic<T>, ic<Q0>) const {
#ifdef IDN_TIMING_NO_ROW_STORES   // timing-only experiment (wrong results): what do the row stores cost?
        return;
#endif
        static_for<2>([&](auto I) {
            constexpr int q = Q0 + decltype(I)::value;
            *reinterpret_cast<f32x4*>(row + 32 * T + 8 * q) = f32x4{O[T][4 * q], O[T][4 * q + 1], O[T][4 * q + 2], O[T][4 * q + 3]};
        });
    }
    template <int T>
    __device__ __forceinline__ void whole(ic<T>) const {
        signs_relu(ic<T>{});
        store2(ic<T>{}, ic<0>{});
        store2(ic<T>{}, ic<2>{});
    }
    template <int T, int Q>
    __device__ __forceinline__ void store1(ic<T>, ic<Q>) const {
#ifdef IDN_TIMING_NO_ROW_STORES
        return;
#endif
#ifdef IDN_TIMING_LINEAR_ROW_STORES   // timing-only (wrong layout): the same bytes into the same 32 rows, 1 KiB contiguous per instruction
        *reinterpret_cast<f32x4*>(lin + (T * 4 + Q) * 256) = f32x4{O[T][4 * Q], O[T][4 * Q + 1], O[T][4 * Q + 2], O[T][4 * Q + 3]};
        return;
#endif
        *reinterpret_cast<f32x4*>(row + 32 * T + 8 * Q) = f32x4{O[T][4 * Q], O[T][4 * Q + 1], O[T][4 * Q + 2], O[T][4 * Q + 3]};
    }
};
// A 256 x 256 layer's tile is exactly one slice of the stream (16 k-steps x 3 fragments), and a wave issues its pieces of
// the next slice in the FIRST half of a slice (two at each of k-steps 0..5).  Tile t - 1 is therefore recorded in the SECOND half of tile t (sign bits
// and ReLU at step 7, one row store at each of steps 8, 10, 12, 14): those four stores are younger than the pieces the barrier at
// the end of the tile waits for, and vmcnt retires in issue order, so that barrier waits `vmcnt(4)` and the stores get
// the next one and a half slices to reach memory, spread over the layer instead of one burst of 32 at its end.
struct RecordInShadow : RecordSide {
    template <int T, int S>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>) const {
        if constexpr (T > 0) {
            if constexpr (S == 7) signs_relu(ic<T - 1>{});
            // one store at each of steps 8, 10, 12, 14: the four waves of a CU run in lockstep, and 4 KiB of stores per k-step
            // is twice what the chip's write path takes from a CU (at 8, 9, 10, 11: forward 4.73 -> 4.63 ms)
            if constexpr (S >= 8 && S < 16 && (S & 1) == 0) store1(ic<T - 1>{}, ic<(S - 8) / 2>{});
        }
    }
};

// End of a layer: the NT accumulator tiles are split into the piece registers (the next layer's input), ReLU'd on
// the way unless the training path has already done that in place.
template <int NT, bool RELU>
__device__ __forceinline__ void convert_layer(const f32x16* O, PTile6* P) {
    static_for<NT>([&](auto T) { convert_tile<RELU>(O[decltype(T)::value], P[decltype(T)::value]); });
}

template <int MODE, bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_bf16x6_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_s = reinterpret_cast<float*>(smem + kX6RingFrags * kFragBytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;

    for (int i = tid; i < kBiasFloats; i += 256) bias_s[i] = a.bias[i];
    __syncthreads();  // the bias block is read (by other waves) before the first slice barrier

    Diag dg;
    WStream6 ws;
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
        PTile6 Pt[8];
        f32x16 O[8];
        auto tiles = [&](auto Q, auto S_) {
            constexpr int q = decltype(Q)::value, s = decltype(S_)::value;
            return Pt[s >> 1].p[q][s & 1];
        };
        uint32_t mk[4] = {0u, 0u, 0u, 0u};   // SAVE: the current layer's ReLU mask bits
        // One hidden layer: MFMAs (training: with the finished tiles recorded in their shadow), then the accumulators
        // become the next layer's input pieces.  `idx` = the layer's matrix in the activation slab (LD floats per row).
        // (always_inline: left to its heuristics hipcc made the inference variant's eight identical calls a real function,
        //  with the 320 registers of O and Pt passed through scratch memory: 11x slower)
        // Training: how many vector-memory operations are younger than a wave's pieces of the slice a layer opens FIRST:
        // after a layer recorded in the shadow (above) the 4 row stores of tile 6 + the 4 of tile 7 (+ the mask store); after a
        // layer recorded at its end 32 + 1.  Never more than were issued (the count must not reach back into the pieces): 8
        // serves both, and the one code instance of layers 1..4 follows both kinds.
#ifndef IDN_TEST_OPEN_YOUNGER        // (overridden only by the audit tool's own negative test: tests/test_boundary_cpu.py)
#define IDN_TEST_OPEN_YOUNGER 8
#define IDN_TEST_MID_YOUNGER 4
#endif
        constexpr int kOpenYounger = IDN_TEST_OPEN_YOUNGER, kMidYounger = IDN_TEST_MID_YOUNGER;
        auto layer = [&](auto F0c, auto NTc, auto KSc, auto LDc, const float* bias_l, auto&& bget, int idx, auto&& hook) __attribute__((always_inline)) {
            constexpr int F0 = decltype(F0c)::value, NT = decltype(NTc)::value, KS = decltype(KSc)::value, LD = decltype(LDc)::value;
            constexpr bool tile_is_slice = NT == 8 && KS == 16 && F0 % kX6SliceFrags == 0;
            if constexpr (SAVE) {
                float* row = a.acts + (long)act_off(idx) * a.p_pad + P * LD + 4 * h;
                if constexpr (tile_is_slice) {
                    const RecordInShadow rec{{O, mk, row, row - (P - (P & ~31L)) * LD - 4 * h + 4 * lane}};
                    run_layer<F0, NT, KS, kOpenYounger, kMidYounger>(O, bias_l, bget, ws, fr, pref, rec, hook);
                    DIAG_BEGIN(dg);
                    rec.whole(ic<NT - 1>{});
                } else {
                    const RecordSide rec{O, mk, row, row};
                    run_layer<F0, NT, KS, (F0 > 0 ? kOpenYounger : 0)>(O, bias_l, bget, ws, fr, pref, NoTileSide{}, hook);
                    DIAG_BEGIN(dg);
                    static_for<NT>([&](auto T) { rec.whole(T); });
                }
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4* mp = reinterpret_cast<u32x4*>(a.acts + (size_t)kActCols * a.p_pad) + mask_index(idx - kActA1, a.p_pad, tile * 4 + wave, lane);
                *mp = u32x4{mk[0], mk[1], mk[2], mk[3]};
                convert_layer<NT, false>(O, Pt);
                DIAG_END(dg, kDgBoundary);
            } else {
                run_layer<F0, NT, KS>(O, bias_l, bget, ws, fr, pref, NoTileSide{}, hook);
                DIAG_BEGIN(dg);
                convert_layer<NT, true>(O, Pt);
                DIAG_END(dg, kDgBoundary);
            }
        };
        // ---- pts_linears.0 : PE(64) -> 256
        layer(ic<f0(0)>{}, ic<8>{}, ic<4>{}, ic<256>{}, bias_h + bias_off(0),
              [&](auto Q, auto S_) { return pe_p[decltype(Q)::value][decltype(S_)::value]; }, kActA1 + 0, NoHook{});
        // ---- pts_linears.1..4 : one code instance (a 256 x 256 layer is four ring lengths of the stream)
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) layer(ic<f0(1)>{}, ic<8>{}, ic<16>{}, ic<256>{}, bias_h + l * 256, tiles, kActA1 + l, NoHook{});
        // ---- pts_linears.5 : [PE(64) | 256] -> 256
        layer(ic<f0(5)>{}, ic<8>{}, ic<20>{}, ic<256>{}, bias_h + bias_off(5),
              [&](auto Q, auto S_) {
                  constexpr int q = decltype(Q)::value, s = decltype(S_)::value;
                  if constexpr (s < 4) return pe_p[q][s];
                  else return Pt[(s - 4) >> 1].p[q][(s - 4) & 1];
              },
              kActA1 + 5, [&]() { load_point<MODE>(a, tile + gridDim.x, wave, m, nxt); });
        // ---- pts_linears.6, .7
        layer(ic<f0(6)>{}, ic<8>{}, ic<16>{}, ic<256>{}, bias_h + bias_off(6), tiles, kActA1 + 6, [&]() { touch_point(nxt); });
        layer(ic<f0(7)>{}, ic<8>{}, ic<16>{}, ic<256>{}, bias_h + bias_off(7), tiles, kActA1 + 7, NoHook{});
        // ---- views_linears.0 (+ alpha_linear as channel 128): [256 | dirPE(32)] -> 160.
        //      Tiles 0..3 are hidden units (recorded, converted); tile 4 is neither: its row 0 is sigma.
        float sigma;
        {
            auto bget8 = [&](auto Q, auto S_) {
                constexpr int q = decltype(Q)::value, s = decltype(S_)::value;
                if constexpr (s < 16) return Pt[s >> 1].p[q][s & 1];
                else return pd_p[q][s - 16];
            };
            if constexpr (SAVE) {
                const RecordSide rec{O, mk, a.acts + (long)act_off(kActV1) * a.p_pad + P * 128 + 4 * h, nullptr};
                run_layer<f0(8), 5, 18, kOpenYounger>(O, bias_h + bias_off(8), bget8, ws, fr, pref);
                static_for<4>([&](auto T) { rec.whole(T); });
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4* mp = reinterpret_cast<u32x4*>(a.acts + (size_t)kActCols * a.p_pad) + mask_index(kActV1 - kActA1, a.p_pad, tile * 4 + wave, lane);
                *mp = u32x4{mk[0], mk[1], mk[2], mk[3]};
                sigma = O[4][0];  // channel 128 = tile 4, register 0, lane half 0
                convert_layer<4, false>(O, Pt);
            } else {
                run_layer<f0(8), 5, 18>(O, bias_h + bias_off(8), bget8, ws, fr, pref);
                sigma = O[4][0];
                convert_layer<4, true>(O, Pt);
            }
        }
        // ---- views_linears.1, .2 : 128 -> 128
        layer(ic<f0(9)>{}, ic<4>{}, ic<8>{}, ic<128>{}, bias_h + bias_off(9), tiles, kActV1 + 1, NoHook{});
        layer(ic<f0(10)>{}, ic<4>{}, ic<8>{}, ic<128>{}, bias_h + bias_off(10), tiles, kActV1 + 2, NoHook{});
        // ---- rgb_linear : 128 -> 3 (rows 0..2 of one tile)
        run_layer<f0(11), 1, 8>(O, bias_h + bias_off(11), tiles, ws, fr, pref);
        finish_pass6<kX6UsedFrags, kX6StreamFrags>(ws);

        if (valid && h == 0) {
            f32x4 o;
            o.x = O[0][0];
            o.y = O[0][1];
            o.z = O[0][2];
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
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModeRays, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, x6::kMlpLds6));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModeRays, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, x6::kMlpLds6));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModeX, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, x6::kMlpLds6));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::mlp_bf16x6_kernel<kModePts, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, x6::kMlpLds6));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ntiles = (n_points + 127) / 128;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    MlpArgs a{packed, folded, x, rays, z, pts, dirs, (long)n_points, n_samples, raw, acts, (long)p_pad};
    ProfScope prof(s, n_points, acts ? IDN_PROF_MLP_FWD_SAVE_X6 : IDN_PROF_MLP_FWD);
    if (x)
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModeX, false>), dim3(grid), dim3(256), x6::kMlpLds6, s, a);
    else if (pts)
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModePts, false>), dim3(grid), dim3(256), x6::kMlpLds6, s, a);
    else if (acts)
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModeRays, true>), dim3(grid), dim3(256), x6::kMlpLds6, s, a);
    else
        hipLaunchKernelGGL((x6::mlp_bf16x6_kernel<kModeRays, false>), dim3(grid), dim3(256), x6::kMlpLds6, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
