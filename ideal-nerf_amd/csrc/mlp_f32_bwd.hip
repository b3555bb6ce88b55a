// Fused backward delta chain of FaceNeRF for the training step, fp32 MFMA (gfx950).
//
// Replaces the eleven per-layer "delta GEMMs" of the backward pass
// (delta_{l-1} = (delta_l . W_l) (.) [a_{l-1} > 0]; loss.backward() through models/face_nerf.py:57-75,
// NeRFs/HeadNeRF/train/audio_exp_nerf.py:534-552) by one kernel built like the forward one
// (mlp_f32.hip): a wave owns 32 points, the TRANSPOSED weights are the streamed A operand, the
// delta of one layer is the accumulator tile set that becomes the B operand of the next, and the
// only HBM traffic per layer is the mask (the saved post-ReLU activation, read in accumulator
// layout) and the delta itself, written row-major for the weight-gradient GEMMs (train.hip).
// The separate GEMMs moved three P x 256 matrices per layer and sat on the HBM/MFMA ridge.
#include "mlp_x6.h"

namespace idn {

__device__ __forceinline__ f32x16 mfma_b(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// transposed stream: fragment (stage, tile t, k-group g), lane (i, h), element j =
//   W_stage[k = 8g + 4h + j][col0 + n = 32t + i]      (zero beyond the valid rows / columns;
//   stage 3 reads alpha_linear's weight as row 128)
// ---------------------------------------------------------------------------
struct BwdPackStage {
    const float* w;      // forward weight [rows = out channels, ld]
    const float* extra;  // optional extra row (alpha_linear) at k == extra_at
    int ld, col0, rows, cols, extra_at;
};
struct BwdPackDesc {
    BwdPackStage st[kBwdStages];
};
__global__ void pack_f32_bwd_kernel(BwdPackDesc d, float4* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= kBwdStreamFrags * 64) return;
    const int f = gid >> 6, lane = gid & 63;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int s = -1;
    for (int i = 0; i < kBwdStages; ++i)
        if (f >= bwd_f0(i) && f < bwd_f0(i) + kBwdNT[i] * kBwdKG[i]) s = i;
    if (s >= 0) {
        const BwdPackStage& S = d.st[s];
        const int rel = f - bwd_f0(s), t = rel / kBwdKG[s], g = rel - t * kBwdKG[s];
        const int n = 32 * t + (lane & 31);
        for (int j = 0; j < 4; ++j) {
            const int k = 8 * g + 4 * (lane >> 5) + j;
            if (n < S.cols) {
                if (k < S.rows) v[j] = S.w[(long)k * S.ld + S.col0 + n];
                else if (S.extra && k == S.extra_at) v[j] = S.extra[n];
            }
        }
    }
    out[gid] = make_float4(v[0], v[1], v[2], v[3]);
}

static void fill_bwd_desc(const idn_facenerf_params& p, BwdPackDesc& d) {
    const int C = p.dim_aud + p.dim_expr + p.dim_latent;
    d.st[0] = {p.rgb_w, nullptr, IDN_W / 2, 0, 3, IDN_W / 2, -1};
    d.st[1] = {p.views_w[2], nullptr, IDN_W / 2, 0, IDN_W / 2, IDN_W / 2, -1};
    d.st[2] = {p.views_w[1], nullptr, IDN_W / 2, 0, IDN_W / 2, IDN_W / 2, -1};
    d.st[3] = {p.views_w[0], p.alpha_w, IDN_W + IDN_VIEWS_CH + p.dim_expr, 0, IDN_W / 2, IDN_W, kSigmaChannel};
    for (int l = 7; l >= 1; --l) {
        BwdPackStage& S = d.st[4 + (7 - l)];
        S = {p.pts_w[l], nullptr, l == 5 ? IDN_PTS_CH + C + IDN_W : IDN_W, l == 5 ? IDN_PTS_CH + C : 0, IDN_W, IDN_W, -1};
    }
}

int launch_pack_f32_bwd(const idn_facenerf_params& p, float* packed_bwd, hipStream_t s) {
    BwdPackDesc d;
    fill_bwd_desc(p, d);
    const int total = kBwdStreamFrags * 64;
    hipLaunchKernelGGL(pack_f32_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d, reinterpret_cast<float4*>(packed_bwd));
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

// ---------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------
struct DeltaArgs {
    const float* wstream;
    const float* acts;
    long p_pad;
    const float* d_rgb;   // [p_pad, 64]
    float* dv0;           // [p_pad, 256]
    float* dv2;           // columns 0..127 of a [p_pad, 256] matrix (row pitch 256): delta of views_linears.2
    float* dv1;           // columns 128..255 of the same matrix: delta of views_linears.1
    float* da[8];
};

// One stage's MFMAs (the forward's pair-step driver with this stream's fragment positions).
template <int F0, int NT, int KG, bool LAST, class BGet, class Side>
__device__ __forceinline__ void run_stage(f32x16 (&out)[NT], BGet&& bget, WStream& ws, FragReader& fr, Side&& side) {
    constexpr int STEPS = KG / 2, NP = NT * STEPS;
    static_assert(KG % 2 == 0 && F0 % 2 == 0, "fragments are consumed in pairs");
    if constexpr (F0 % kSliceFrags == 0) {
        ws.open_slice();
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
        side(ic<t>{}, ic<s>{}, ic<0>{});
        out[t] = mfma_b(a0.x, bget(ic<g0>{}, ic<0>{}), out[t]);
        out[t] = mfma_b(a0.y, bget(ic<g0>{}, ic<1>{}), out[t]);
        out[t] = mfma_b(a0.z, bget(ic<g0>{}, ic<2>{}), out[t]);
        out[t] = mfma_b(a0.w, bget(ic<g0>{}, ic<3>{}), out[t]);
        out[t] = mfma_b(a1.x, bget(ic<g1>{}, ic<0>{}), out[t]);
        out[t] = mfma_b(a1.y, bget(ic<g1>{}, ic<1>{}), out[t]);
        out[t] = mfma_b(a1.z, bget(ic<g1>{}, ic<2>{}), out[t]);
        out[t] = mfma_b(a1.w, bget(ic<g1>{}, ic<3>{}), out[t]);
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

// In the shadow of tile t's MFMAs (pair-steps 0 ..): the finished tile t-1 is masked (step 1) with the
// layer's ReLU bits (one uint4 per lane and layer, written by the forward in this very accumulator layout and
// loaded a whole stage ahead; gathering the saved activations instead cost four 16-byte loads per tile and most
// of the kernel's parked cycles), scattered into the wave's LDS patch (step 2) and written
// out as row segments: the four row values of quad q are READ from the patch at step 3 + q and
// STORED at step 4 + q.  The patch reads are inline asm like the fragment reads: LDS returns in
// order, so the pair-step's counted `lgkmcnt(2)` (FragReader::retire) has covered them by the time
// they are stored, and the compiler never waits `lgkmcnt(0)` for them (it did, sixteen exposed LDS
// round trips per tile, when these were plain loads).  The last tile of a stage, and every tile
// of a stage too short for this schedule, is finished after the stage (finish()).
template <int NT, int STEPS, int LD>
struct MaskSide {
    static constexpr bool kShadowStore = STEPS >= 8;
    f32x16* out;
    const uint32_t* mk;        // [4] mask bits of this layer: dword k = tiles 2k, 2k+1, value i of the pair at bit 31 - i
    __amdgpu_buffer_rsrc_t rsrc;  // this wave's 32 rows of the delta matrix (LD floats per row)
    uint32_t voff;             // byte offset of [row h][column m]
    float* stage;              // this wave's 32 x 33 transpose patch
    float* rb;                 // [4] row values in flight between their read and their store
    uint32_t raddr;            // LDS byte address of patch[h][m]
    int m, h;
    template <int T>
    __device__ __forceinline__ void apply(ic<T>) const {
        const uint32_t w = mk[T >> 1];
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value, i = 16 * (T & 1) + r;
            const int off = (int)(w << i) >> 31;   // all ones where the unit was off (its pre-activation <= 0)
            out[T][r] = __uint_as_float(__float_as_uint(out[T][r]) & ~(uint32_t)off);
        });
    }
    template <int T>
    __device__ __forceinline__ void scatter(ic<T>) const {
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value;
            stage[m * kStagePitch + (r & 3) + 8 * (r >> 2) + 4 * h] = out[T][r];
        });
        asm volatile("" ::: "memory");   // the asm reads below come after these writes (LDS executes a wave's operations in order)
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
    // after the stage: whatever the shadow schedule did not cover
    __device__ __forceinline__ void finish() const {
        apply(ic<NT - 1>{});
        if constexpr (kShadowStore) flush_tile(ic<NT - 1>{});
        else static_for<NT>([&](auto T) { flush_tile(T); });
    }
    template <int T, int S, int H>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>, ic<H>) const {
        constexpr int kApplyStep = STEPS > 1 ? 1 : 0;   // a one-step stage (rgb_linear^T) applies before it reloads
        if constexpr (S == kApplyStep && T > 0) apply(ic<T - 1>{});
        if constexpr (kShadowStore && T > 0) {
            if constexpr (S == 2) scatter(ic<T - 1>{});
            if constexpr (S >= 4 && S <= 7) rows_store(ic<T - 1>{}, ic<S - 4>{});
            if constexpr (S >= 3 && S <= 6) rows_read(ic<S - 3>{});
        }
    }
};

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&t)[NT]) {
    static_for<NT>([&](auto T) {
        static_for<16>([&](auto R) { t[decltype(T)::value][decltype(R)::value] = 0.0f; });
    });
}

constexpr int kDeltaLds = kRingFrags * kFragBytes + 4 * kStageFloats * 4;

__global__ __launch_bounds__(256, 1) void delta_chain_kernel(DeltaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem + kRingFrags * kFragBytes) + wave * kStageFloats;

    Diag dg;
    WStream ws;
    ws.dg = &dg;
    ws.init(a.wstream, kBwdNumSlices, ring, tid, wave);
    FragReader fr;
    fr.addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    fr.addr1 = fr.addr0 + 64 * kFragBytes;
    const long ntiles = a.p_pad >> 7;
    const uint32_t raddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)stage + (h * kStagePitch + m) * 4;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p0 = tile * 128 + wave * 32;
        const long P = p0 + m;
        // d raw of this lane's point: rgb in k-channels 0..2 (lane half 0), sigma in k-channel 128
        const f32x4 drgb = *reinterpret_cast<const f32x4*>(a.d_rgb + P * 64);
        const float dsig = a.dv0[P * 256 + kSigmaChannel];

        f32x16 A[8], B[8];
        float rb[4];
        // ReLU masks: mask_use = this stage's bits, mask_nxt = the next stage's, in flight since the previous stage began
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* mbase = reinterpret_cast<const u32x4*>(a.acts + (size_t)kActCols * a.p_pad);
        auto mask_load = [&](int id) { return mbase[mask_index(id, a.p_pad, tile * 4 + wave, lane)]; };
        u32x4 mask_nxt = mask_load(10);
        auto tiles_get = [](f32x16* arr) {
            return [arr](auto G, auto J) {
                constexpr int g = decltype(G)::value, j = decltype(J)::value;
                return arr[g >> 2][(g & 3) * 4 + j];
            };
        };
        // one stage: zero accumulators, MFMAs with the mask in their shadow, last tile's mask, store
        // `next_id`: the mask layer of the stage after this one (-1: none), loaded now and used then
        auto stage_run = [&](auto Sc, auto F0c, auto LASTc, auto LDc, auto& out, auto&& bget, int next_id, float* dst) {
            constexpr int S = decltype(Sc)::value, F0 = decltype(F0c)::value;
            constexpr int NT = kBwdNT[S], KG = kBwdKG[S];
            zero_tiles<NT>(out);
            constexpr int LD = decltype(LDc)::value;
            const u32x4 mv = mask_nxt;
            const uint32_t mk[4] = {mv.x, mv.y, mv.z, mv.w};
            if (next_id >= 0) mask_nxt = mask_load(next_id);
            const MaskSide<NT, KG / 2, LD> side{&out[0], mk, rows_rsrc(dst + p0 * LD, LD), (uint32_t)((h * LD + m) * 4), stage, rb, raddr, m, h};
            run_stage<F0, NT, KG, decltype(LASTc)::value != 0>(out, bget, ws, fr, side);
            side.finish();
        };
        f32x16(&A4)[4] = reinterpret_cast<f32x16(&)[4]>(A);
        f32x16(&B4)[4] = reinterpret_cast<f32x16(&)[4]>(B);

        // 0: rgb_linear^T : d rgb (3) -> delta of views_linears.2, masked by its output v3
        stage_run(ic<0>{}, ic<bwd_f0(0)>{}, ic<0>{}, ic<256>{}, A4,
                  [&](auto G, auto J) {
                      constexpr int g = decltype(G)::value, j = decltype(J)::value;
                      if constexpr (g == 0 && j < 3) return h ? 0.0f : drgb[j];
                      else return 0.0f;
                  },
                  9, a.dv2);
        // 1: views_linears.2^T -> delta of views_linears.1 (mask v2);  2: views_linears.1^T -> views_linears.0 (mask v1)
        stage_run(ic<1>{}, ic<bwd_f0(1)>{}, ic<0>{}, ic<256>{}, B4, tiles_get(A), 8, a.dv1);
        stage_run(ic<2>{}, ic<bwd_f0(2)>{}, ic<0>{}, ic<256>{}, A4, tiles_get(B), 7, a.dv0);
        // 3: views_linears.0[:, :256]^T + alpha_linear^T (d sigma as k-channel 128) -> delta of pts_linears.7 (mask a8)
        stage_run(ic<3>{}, ic<bwd_f0(3)>{}, ic<1>{}, ic<256>{}, B,
                  [&](auto G, auto J) {
                      constexpr int g = decltype(G)::value, j = decltype(J)::value;
                      if constexpr (g < 16) return A[g >> 2][(g & 3) * 4 + j];
                      else if constexpr (g == 16 && j == 0) return h ? 0.0f : dsig;
                      else return 0.0f;
                  },
                  6, a.da[7]);
        finish_pass<kBwdHeadFrags, kBwdTrunk0>(ws);   // walk the padding up to the trunk stages
        // 4..9: pts_linears.7 .. .2 ^T in pairs (B -> A -> B), then pts_linears.1^T (last of the pass)
#pragma unroll 1
        for (int l = 7; l >= 3; l -= 2) {
            stage_run(ic<4>{}, ic<bwd_f0(4)>{}, ic<0>{}, ic<256>{}, A, tiles_get(B), l - 2, a.da[l - 1]);   // masks a_l: id l - 1; next: l - 2
            stage_run(ic<5>{}, ic<bwd_f0(5)>{}, ic<0>{}, ic<256>{}, B, tiles_get(A), l - 3, a.da[l - 2]);   // masks a_(l-1): id l - 2; next: l - 3
        }
        stage_run(ic<10>{}, ic<bwd_f0(10)>{}, ic<1>{}, ic<256>{}, A, tiles_get(B), -1, a.da[0]);   // masks a1: id 0
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}

int launch_delta_chain(const float* packed_bwd, const float* acts, int64_t p_pad, const float* d_rgb, float* dv0,
                       float* dv2, float* dv1, float* const da[8], hipStream_t s) {
    if (p_pad <= 0) return IDN_OK;
    if (p_pad % 128) return fail(IDN_EINVAL, "delta chain: p_pad %lld is not a multiple of 128", (long long)p_pad);
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&delta_chain_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kDeltaLds));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ntiles = p_pad / 128;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    DeltaArgs a{packed_bwd, acts, (long)p_pad, d_rgb, dv0, dv2, dv1, {}};
    for (int l = 0; l < 8; ++l) a.da[l] = da[l];
    ProfScope prof(s, p_pad, IDN_PROF_DELTA_CHAIN);
    hipLaunchKernelGGL(delta_chain_kernel, dim3(grid), dim3(256), kDeltaLds, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}


// ===========================================================================================
// The same chain on the bf16 matrix pipe, fp32-grade: six bf16 piece products per fp32 product (mlp_x6.h; the
// forward in mlp_bf16x6.hip, the weight-gradient GEMMs in train.hip).  A stage's input delta is held as pieces
// (192 registers for 256 channels x 32 points), its output as the fp32 accumulator tiles; at the end of a stage
// the accumulators are masked, written to the delta matrix (this lane's row: a quad of registers is 16 contiguous
// bytes of it) and split into the piece registers, which the finished stage no longer needs.
// Stream: per (tile, 16-channel k-step) a triple of fragments (p1, p2, p3) in 48-fragment ring slots (idn_internal.h,
// mlp_x6.h): a slice is 16 k-steps as in the fp32 stream (16 pairs), so every stage keeps its ring phase; 3.19 MiB.
// ===========================================================================================
constexpr int bwd6_f0(int s) { return kX6KFrags * (bwd_f0(s) / 2); }   // bwd_f0 counts two fragments per k-step
constexpr int kBwd6StreamFrags = kX6KFrags * (kBwdStreamFrags / 2);    // 3264
constexpr int kBwd6NumSlices = kBwd6StreamFrags / kX6SliceFrags;       // 68
static_assert(kBwd6StreamFrags % kX6SliceFrags == 0 && kBwd6NumSlices % kRingSlots == 0 && bwd6_f0(4) % kX6RingFrags == 0, "ring phase of the trunk stages");

__global__ void pack_bf16x6_bwd_kernel(BwdPackDesc d, uint4* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= kBwd6StreamFrags * 64) return;
    const int f = gid >> 6, lane = gid & 63, part = f % kX6KFrags;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    int s = -1;
    for (int i = 0; i < kBwdStages; ++i)
        if (f >= bwd6_f0(i) && f < bwd6_f0(i) + kX6KFrags * kBwdNT[i] * (kBwdKG[i] / 2)) s = i;
    if (s >= 0) {
        const BwdPackStage& S = d.st[s];
        const int ksn = kBwdKG[s] / 2;
        const int rel = (f - bwd6_f0(s)) / kX6KFrags, t = rel / ksn, ks = rel - t * ksn;
        const int n = 32 * t + (lane & 31), h = lane >> 5;
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + (j & 3) + 8 * (j >> 2) + 4 * h;   // the element order an accumulator tile splits into
            float v = 0.f;
            if (n < S.cols) {
                if (k < S.rows) v = S.w[(long)k * S.ld + S.col0 + n];
                else if (S.extra && k == S.extra_at) v = S.extra[n];
            }
            const unsigned u1 = __float_as_uint(v);
            const unsigned p1 = (u1 + 0x7fffu + ((u1 >> 16) & 1u)) >> 16;   // bf16, round to nearest even (finite weights)
            const float r1 = v - __uint_as_float(p1 << 16);
            const unsigned u2 = __float_as_uint(r1);
            const unsigned p2 = (u2 + 0x7fffu + ((u2 >> 16) & 1u)) >> 16;
            const float r2 = r1 - __uint_as_float(p2 << 16);
            const unsigned u3 = __float_as_uint(r2);
            const unsigned p3 = (u3 + 0x7fffu + ((u3 >> 16) & 1u)) >> 16;
            const unsigned bits = part == 0 ? p1 : (part == 1 ? p2 : p3);
            w[j >> 1] |= bits << (16 * (j & 1));
        }
    }
    out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

size_t bwd_stream_floats_x6() { return (size_t)kBwd6StreamFrags * kFragFloats; }

int launch_pack_bf16x6_bwd(const idn_facenerf_params& p, float* packed_bwd, hipStream_t s) {
    BwdPackDesc d;
    fill_bwd_desc(p, d);
    const int total = kBwd6StreamFrags * 64;
    hipLaunchKernelGGL(pack_bf16x6_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d, reinterpret_cast<uint4*>(packed_bwd));
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

namespace x6 {

// One stage's MFMAs: zeroed accumulators, KS k-steps of six piece products per tile (run_layer of mlp_bf16x6.hip
// without the bias).  LAST: nothing is read ahead past this stage (the padding before the trunk / the end of the pass).
template <int F0, int NT, int KS, bool LAST, int OPEN_YOUNGER, int MID_YOUNGER, class BGet, class Side>
__device__ __forceinline__ void run_stage6(f32x16* O, BGet&& bget, WStream6& ws, FragReader& fr, f32x4 (&pref)[3], Side&& side) {
    constexpr int NP = NT * KS;
    static_assert(F0 % kX6KFrags == 0, "triples");
    static_for<NT>([&](auto T) { static_for<16>([&](auto R) { O[decltype(T)::value][decltype(R)::value] = 0.0f; }); });
    if constexpr (F0 % kX6SliceFrags == 0) {
        // OPEN_YOUNGER: the stage before this one has issued that many row stores after this wave's last piece of the slice
        // being opened: they stay in flight.  (Never more than were issued: the count must not reach back into the pieces --
        // tools/audit_asm_loads.py counts them on every path.)
        ws.template open_slice<OPEN_YOUNGER>();
        static_for<3>([&](auto Q) { pref[decltype(Q)::value] = issue6<F0 + decltype(Q)::value>(fr); });
        retire3<0>(pref);
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
        O[t] = mfma_bf(a[0], b1, O[t]);
        O[t] = mfma_bf(a[0], b2, O[t]);
        O[t] = mfma_bf(a[1], b1, O[t]);
        O[t] = mfma_bf(a[1], b2, O[t]);
        side(ic<t>{}, ic<s>{});   // the finished tile t - 1 is masked and written out in this tile's shadow
        O[t] = mfma_bf(a[0], b3, O[t]);
        O[t] = mfma_bf(a[2], b1, O[t]);
        if constexpr (next_crosses && pi + 1 < NP) {
            // MID_YOUNGER (trunk stages, whose tiles are slices): the row stores issued in the second half of the slice that
            // ends here -- none in tile 0 -- are younger than the pieces of the slice being opened
            ws.template open_slice<(t >= 1 ? MID_YOUNGER : 0)>();
            static_for<3>([&](auto Q) { n[decltype(Q)::value] = issue6<f + kX6KFrags + decltype(Q)::value>(fr); });
        }
        a[0] = n[0];
        a[1] = n[1];
        a[2] = n[2];
    });
    if constexpr (!LAST && (F0 + kX6KFrags * NP) % kX6SliceFrags != 0) retire3<0>(a);   // hand over retired fragments
    pref[0] = a[0];
    pref[1] = a[1];
    pref[2] = a[2];
}

// A tile of a stage, once its accumulator is complete: the ReLU bits of the layer it is the delta of are applied and the
// 16 values go to this lane's row of the delta matrix.  Stores count in vmcnt on gfx9, vmcnt retires in issue order, and
// every slice barrier waits for the wave's pieces of the next slice, so WHEN a store is issued matters (mlp_bf16x6.hip,
// RecordSide / RecordInShadow): head stages do it after the stage, all tiles before any is converted; trunk stages, whose
// tile is one slice, in the second half of the next tile (operator() below).
template <int KS>
struct MaskStoreSide {
    f32x16* O;
    const uint32_t* mk;   // dword k = tiles 2k, 2k+1, value i of the pair at bit 31 - i
    float* row;           // this lane's row of the delta matrix + 4 h
    float* lin;           // timing-only experiment: first float of this wave's 32 rows + 4 lane
    template <int T>
    __device__ __forceinline__ void apply(ic<T>) const {
        const uint32_t w = mk[T >> 1];
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value, i = 16 * (T & 1) + r;
            const int off = (int)(w << i) >> 31;   // all ones where the unit was off (its pre-activation <= 0)
            O[T][r] = __uint_as_float(__float_as_uint(O[T][r]) & ~(uint32_t)off);
        });
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
        apply(ic<T>{});
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
    // A trunk stage's tile is exactly one slice of the stream, and a wave issues its pieces of the next slice in the FIRST
    // half of a slice: tile t - 1 is masked at step 7 of tile t and written at steps 8, 10, 12, 14 -- four stores younger than the
    // pieces the barrier at the end of the tile waits for (`vmcnt(4)` there: vmcnt retires in issue order), with one and a
    // half slices to reach memory, instead of a burst of 32 at the end of the stage.  Other stages: after the stage.
    template <int T, int S>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>) const {
        if constexpr (KS == 16 && T > 0) {
            if constexpr (S == 7) apply(ic<T - 1>{});
            if constexpr (S >= 8 && S < 16 && (S & 1) == 0) store1(ic<T - 1>{}, ic<(S - 8) / 2>{});   // steps 8, 10, 12, 14
        }
    }
};

constexpr int kDelta6Lds = kX6RingFrags * kFragBytes;

__global__ __launch_bounds__(256, 1) void delta_chain_x6_kernel(DeltaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;

    Diag dg;
    WStream6 ws;
    ws.dg = &dg;
    ws.init(a.wstream, kBwd6NumSlices, ring, tid, wave);
    FragReader fr;
    fr.addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    fr.addr1 = fr.addr0 + 64 * kFragBytes;
    const long ntiles = a.p_pad >> 7;
    f32x4 pref[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long P = tile * 128 + wave * 32 + m;
        // d raw of this lane's point: rgb in k-channels 0..2 (lane half 0), sigma in k-channel 128
        const f32x4 drgb = *reinterpret_cast<const f32x4*>(a.d_rgb + P * 64);
        const float dsig = a.dv0[P * 256 + kSigmaChannel];
        // as pieces of one k-step: word 0 = channels (0, 1), word 1 = (2, 3) of lane half 0; everything else zero
        f32x4 in_rgb[3], in_sig[3];
        {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            float r[3][2], g[3];
            split3(h ? 0.0f : drgb[0], h ? 0.0f : drgb[1], r[0][0], r[1][0], r[2][0]);
            split3(h ? 0.0f : drgb[2], 0.0f, r[0][1], r[1][1], r[2][1]);
            float unused0, unused1, unused2;
            (void)unused0; (void)unused1; (void)unused2;
            split3(h ? 0.0f : dsig, 0.0f, g[0], g[1], g[2]);
            static_for<3>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                in_rgb[q] = z4;
                in_rgb[q][0] = r[q][0];
                in_rgb[q][1] = r[q][1];
                in_sig[q] = z4;
                in_sig[q][0] = g[q];
            });
        }

        settle(in_rgb);   // (asm conversion results, read by the first stage's MFMAs right away: mlp_x6.h)
        settle(in_sig);
        PTile6 Pt[8];
        f32x16 O[8];
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* mbase = reinterpret_cast<const u32x4*>(a.acts + (size_t)kActCols * a.p_pad);
        auto mask_load = [&](int id) { return mbase[mask_index(id, a.p_pad, tile * 4 + wave, lane)]; };
        u32x4 mask_nxt = mask_load(10);
        auto tiles = [&](auto Q, auto S_) {
            constexpr int q = decltype(Q)::value, s = decltype(S_)::value;
            return Pt[s >> 1].p[q][s & 1];
        };
        // one stage: MFMAs, then mask (this stage's ReLU bits, loaded a stage ahead), store the delta rows, split into pieces
        auto stage_run = [&](auto F0c, auto NTc, auto KSc, auto LASTc, auto LDc, auto YOUNGERc, auto&& bget, int next_id, float* dst) __attribute__((always_inline)) {
            constexpr int NT = decltype(NTc)::value, LD = decltype(LDc)::value;
            const u32x4 mv = mask_nxt;
            const uint32_t mk[4] = {mv.x, mv.y, mv.z, mv.w};
            if (next_id >= 0) mask_nxt = mask_load(next_id);
            const MaskStoreSide<decltype(KSc)::value> side{O, mk, dst + P * LD + 4 * h, dst + (P & ~31L) * LD + 4 * lane};
            // a trunk stage after the first opens its first slice behind the row stores of tiles 6 and 7 of the stage before it
            constexpr bool trunk = decltype(F0c)::value >= bwd6_f0(4);
            run_stage6<decltype(F0c)::value, NT, decltype(KSc)::value, decltype(LASTc)::value != 0, decltype(YOUNGERc)::value, (trunk ? 4 : 0)>(O, bget, ws, fr, pref, side);
            if constexpr (trunk) side.whole(ic<NT - 1>{});
            else static_for<NT>([&](auto T) { side.whole(T); });
            static_for<NT>([&](auto T) { convert_tile<false>(O[decltype(T)::value], Pt[decltype(T)::value]); });
        };
        // 0: rgb_linear^T : d rgb (3) -> delta of views_linears.2, masked by its output v3
        stage_run(ic<bwd6_f0(0)>{}, ic<4>{}, ic<1>{}, ic<0>{}, ic<256>{}, ic<0>{}, [&](auto Q, auto) { return in_rgb[decltype(Q)::value]; }, 9, a.dv2);
        // 1: views_linears.2^T -> delta of views_linears.1 (mask v2);  2: views_linears.1^T -> views_linears.0 (mask v1)
        stage_run(ic<bwd6_f0(1)>{}, ic<4>{}, ic<8>{}, ic<0>{}, ic<256>{}, ic<0>{}, tiles, 8, a.dv1);
        stage_run(ic<bwd6_f0(2)>{}, ic<4>{}, ic<8>{}, ic<0>{}, ic<256>{}, ic<0>{}, tiles, 7, a.dv0);
        // 3: views_linears.0[:, :256]^T + alpha_linear^T (d sigma as k-channel 128) -> delta of pts_linears.7 (mask a8)
        stage_run(ic<bwd6_f0(3)>{}, ic<8>{}, ic<9>{}, ic<1>{}, ic<256>{}, ic<0>{},
                  [&](auto Q, auto S_) {
                      constexpr int q = decltype(Q)::value, s = decltype(S_)::value;
                      if constexpr (s < 8) return Pt[s >> 1].p[q][s & 1];
                      else return in_sig[q];
                  },
                  6, a.da[7]);
        finish_pass6<kX6KFrags * (kBwdHeadFrags / 2), bwd6_f0(4)>(ws);   // walk the padding up to the trunk stages
        // 4: pts_linears.7^T follows the padding walk, not a stage: its first slice is opened with a full wait.
        // 5..9: pts_linears.6 .. .2 ^T, one code instance (a trunk stage is four ring lengths), each opening its first slice
        // behind the 8 row stores of tiles 6 and 7 of the stage before it; then pts_linears.1^T.
        stage_run(ic<bwd6_f0(4)>{}, ic<8>{}, ic<16>{}, ic<0>{}, ic<256>{}, ic<0>{}, tiles, 5, a.da[6]);                   // masks a7: id 6
#pragma unroll 1
        for (int L = 6; L >= 2; --L)
            stage_run(ic<bwd6_f0(5)>{}, ic<8>{}, ic<16>{}, ic<0>{}, ic<256>{}, ic<8>{}, tiles, L - 2, a.da[L - 1]);       // masks a_L: id L - 1
        stage_run(ic<bwd6_f0(10)>{}, ic<8>{}, ic<16>{}, ic<1>{}, ic<256>{}, ic<8>{}, tiles, -1, a.da[0]);                 // masks a1: id 0
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}

}  // namespace x6

int launch_delta_chain_x6(const float* packed_bwd, const float* acts, int64_t p_pad, const float* d_rgb, float* dv0,
                          float* dv2, float* dv1, float* const da[8], hipStream_t s) {
    if (p_pad <= 0) return IDN_OK;
    if (p_pad % 128) return fail(IDN_EINVAL, "delta chain: p_pad %lld is not a multiple of 128", (long long)p_pad);
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x6::delta_chain_x6_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, x6::kDelta6Lds));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ntiles = p_pad / 128;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    DeltaArgs a{packed_bwd, acts, (long)p_pad, d_rgb, dv0, dv2, dv1, {}};
    for (int l = 0; l < 8; ++l) a.da[l] = da[l];
    ProfScope prof(s, p_pad, IDN_PROF_DELTA_CHAIN_X6);
    hipLaunchKernelGGL(x6::delta_chain_x6_kernel, dim3(grid), dim3(256), x6::kDelta6Lds, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
