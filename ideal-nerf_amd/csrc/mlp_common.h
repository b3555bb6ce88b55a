// Shared machinery of the fused PE + MLP kernels (fp32 and bf16x3 variants): compile-time
// loops, the weight stream (global -> LDS ring, piece-wise prefetch), the inline-asm fragment
// reader, bias / ReLU helpers, the exact-phase positional encoding and the kernel arguments.
// gfx950 only.  See DESIGN.md "MLP kernel".
#pragma once
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

// Diagnostic build only (-DIDN_DIAG): per-wave cycle totals by category, written to a buffer
// nothing else reads.  Never compiled into the shipped library (cdna_hip_programming.md 7).
#ifdef IDN_DIAG
#define DIAG_ONLY(x) x
__device__ unsigned long long g_diag[8];
enum { kDgTotal = 0, kDgInput = 1, kDgBarrier = 2, kDgBoundary = 3, kDgStore = 4 };
struct Diag {
    unsigned long long acc[5] = {0, 0, 0, 0, 0};
    unsigned long long t0 = 0;
    __device__ __forceinline__ void begin() { __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void end(int cat) {
        __builtin_amdgcn_sched_barrier(0);
        acc[cat] += clock64() - t0;
        __builtin_amdgcn_sched_barrier(0);
    }
};
#define DIAG_BEGIN(d) (d).begin()
#define DIAG_END(d, c) (d).end(c)
#else
#define DIAG_ONLY(x)
struct Diag {};
#define DIAG_BEGIN(d)
#define DIAG_END(d, c)
#endif

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------
// weight stream: global -> LDS ring
//
// The ring holds kRingSlots slices (idn_internal.h).  A slice is fetched by "pieces": one
// buffer_load_dwordx4 ... lds per wave each, i.e. NW KiB per piece for a workgroup of NW waves.
// While slice s is consumed, the pieces of slice s + kRingSlots-1 are issued, one per
// fragment-pair step, into the slot slice s-1 just vacated.  A slice is opened by a COUNTED wait +
// raw barrier: s_waitcnt vmcnt(N) leaves the N pieces of the younger slices in flight (N = 0 for
// the two-slot ring).  Other vector-memory operations of the wave only make the wait stricter.
//
// Piece form: descriptor (4 SGPRs) + one constant VGPR (lane * 16) + a scalar stream offset.
// Beside an MFMA chain (tools/glds_ubench.hip) it costs the issuing wave about half of
// global_load_lds with per-lane 64-bit pointers and the clock holds higher.  The descriptor's
// num_records bounds every piece to the stream (out of range reads zeros instead of faulting).
// What was tried and made no difference: issuing 2 or 4 pieces per step, staggering the
// workgroups of an XCD, never waiting for the pieces (timing-only) -- DESIGN.md section 3.
// ---------------------------------------------------------------------------
constexpr int kAhead = kRingSlots - 1;  // slices in flight ahead of the consumer

// DUAL: the stream alternates between TWO packed networks pass by pass (the fused ray kernel: passes [0, dual_first) of
// every dual_period passes read stream 0, the others stream 1); the descriptor is switched where the prefetch wraps.
template <int NW, int SLICE_FRAGS = kSliceFrags, int SLOTS = kRingSlots, bool DUAL = false>
struct WStreamT {
    static constexpr int kSliceFragsT = SLICE_FRAGS;               // fragments per ring slot (64; the six-piece streams: 48)
    static constexpr int kSliceBytesT = SLICE_FRAGS * kFragBytes;
    static constexpr int kSlotsT = SLOTS;                          // ring slots: slices are fetched SLOTS - 1 ahead of their use
    static constexpr int kAheadT = SLOTS - 1;
    static constexpr int kPieceBytes = NW * kFragBytes;
    static constexpr int kPieces = kSliceBytesT / kPieceBytes;     // pieces per slice
    static constexpr int kVmcntOpen = (kAheadT - 1) * kPieces;     // younger pieces allowed in flight at a barrier
    static_assert(kSliceBytesT % kPieceBytes == 0 && kPieces % 4 == 0, "a wave's pieces come in groups of four (one M0 / scalar offset per group)");

    Diag* dg;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;      // this lane's 16-byte column inside a piece
    uint32_t soff;      // byte offset of the slice currently being fetched (wave-uniform)
    int next_slice;
    int num_slices;     // slices in this stream (kNumSlices, or kPlainNumSlices for the plain-bf16 stream)
    char* ring_wave;    // ring + this wave's block of a slice (wave-uniform LDS destination base)
    const float* dual_stream[2];          // DUAL only
    int dual_first, dual_period, dual_pass;   // DUAL only: the pass whose slices are being FETCHED

    __device__ __forceinline__ void init(const float* stream, int slices, char* ring, int tid, int wave) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, slices * kSliceBytesT, 0x00020000);  // raw, untyped
        // wave w fetches the kPieces consecutive fragments w * kPieces .. of every slice: four consecutive
        // pieces then differ only in the instruction's immediate offset (1 KiB steps, applied to the global
        // and the LDS address alike), so a slot needs kPieces / 4 M0 / scalar-offset values instead of
        // kPieces -- the per-site constants the compiler keeps in SGPRs (and spills through VALU lanes)
        voff = (tid & 63) * 16 + wave * (kPieces * kFragBytes);
        soff = 0;
        next_slice = 0;
        num_slices = slices;
        ring_wave = ring + wave * (kPieces * kFragBytes);
        static_for<kAheadT>([&](auto S_) { issue_rest<decltype(S_)::value, 0>(); });  // slices 0 .. kAheadT-1
    }
    __device__ __forceinline__ void advance() {
        soff += kSliceBytesT;
        if (++next_slice == num_slices) {
            next_slice = 0;
            soff = 0;
            if constexpr (DUAL) {
                if (++dual_pass == dual_period) dual_pass = 0;
                rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dual_stream[dual_pass < dual_first ? 0 : 1]), 0, num_slices * kSliceBytesT, 0x00020000);
            }
        }
    }
    __device__ __forceinline__ void init_dual(const float* stream0, const float* stream1, int first, int period, int slices, char* ring, int tid, int wave) {
        static_assert(DUAL, "two streams");
        dual_stream[0] = stream0;
        dual_stream[1] = stream1;
        dual_first = first;
        dual_period = period;
        dual_pass = 0;
        init(stream0, slices, ring, tid, wave);
    }
    template <int SLOT, int J>
    __device__ __forceinline__ void issue_piece() {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(ring_wave + SLOT * kSliceBytesT + (J / 4) * (4 * kFragBytes)), 16, voff,
                                                 soff + (J / 4) * (4 * kFragBytes), (J % 4) * kFragBytes, 0);
        if constexpr (J == kPieces - 1) advance();
    }
    template <int SLOT, int J0>
    __device__ __forceinline__ void issue_rest() {
        static_for<kPieces - J0>([&](auto I) { issue_piece<SLOT, J0 + decltype(I)::value>(); });
    }
    // Open the next slice: this wave's pieces of it have landed; after the barrier every wave's
    // have, and every wave is done reading the slot the next pieces will overwrite.
    // YOUNGER: vector-memory operations this wave has issued AFTER its last piece of the slice being opened (the row
    // stores at the end of a training layer).  vmcnt retires in issue order -- loads, LDS-DMA pieces and stores alike --
    // so they may stay in flight: a wait for them here would park every wave behind its own burst of writes.
    template <int YOUNGER = 0>
    __device__ __forceinline__ void open_slice() {
        DIAG_BEGIN(*dg);
        static_assert(kVmcntOpen + YOUNGER <= 63, "vmcnt is a 6-bit field");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kVmcntOpen + YOUNGER) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DIAG_END(*dg, kDgBarrier);
    }
    // the piece (if any) to issue at the pair-step that consumes fragment F
    template <int F>
    __device__ __forceinline__ void step_piece() {
        constexpr int jpos = (F % SLICE_FRAGS) / 2;
        constexpr int slot = (F / SLICE_FRAGS + kAheadT) % SLOTS;
        if constexpr (jpos < kPieces) issue_piece<slot, jpos>();
    }
    // the same by position: piece JPOS (if the slice has that many) of the slice fetched while slice F / SLICE_FRAGS is consumed
    template <int F, int JPOS>
    __device__ __forceinline__ void step_piece_at() {
        constexpr int slot = (F / SLICE_FRAGS + kAheadT) % SLOTS;
        if constexpr (JPOS < kPieces) issue_piece<slot, JPOS>();
    }
};
using WStream = WStreamT<4>;

// ---------------------------------------------------------------------------
// A-fragment reads.  hipcc (ROCm 7.2) waits lgkmcnt(0) after a prefetching ds_read -- i.e.
// for the read it has just issued.  The reads are therefore issued from inline asm, which
// the compiler does not count, and retired by a counted wait tied to the destination
// registers ("+v"): LDS operations return in order, so lgkmcnt(2) right after issuing pair
// p+1 means pair p has landed.  Compiler-issued LDS/SMEM operations in between only make
// these waits stricter (never weaker), and its own counted waits likewise
// (cdna_hip_programming.md section 5.7).
// ---------------------------------------------------------------------------
struct FragReader {
    uint32_t addr0, addr1;  // LDS byte address of this lane's 16 bytes in fragment 0 / fragment 64
    uint32_t addr2;         // ... / fragment 128 (rings longer than 128 fragments: the three-slot six-piece ring)
    f32x4 pref0, pref1;     // fragment pair issued ahead of its consumer (layer / slice start)

    template <int F, int RING_FRAGS = kRingFrags>
    __device__ __forceinline__ f32x4 issue() const {
        constexpr int fr = F % RING_FRAGS;
        f32x4 v;
        if constexpr (fr < 64)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr0), "n"(fr * kFragBytes) : "memory");
        else if constexpr (fr < 128)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr1), "n"((fr - 64) * kFragBytes) : "memory");
        else
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr2), "n"((fr - 128) * kFragBytes) : "memory");
        return v;
    }
    // all but the newest `Newer` LDS reads of this wave have completed => v0, v1 are valid
    template <int Newer>
    static __device__ __forceinline__ void retire(f32x4& v0, f32x4& v1) {
#ifdef IDN_TIMING_NO_FRAG_WAIT   // timing-only experiment (wrong results): is the fragment-read latency exposed?
        asm volatile("" : "+v"(v0), "+v"(v1)::"memory");
        return;
#endif
        if constexpr (Newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v0), "+v"(v1)::"memory");
    }
};

// End of a pass: walk the unused tail of the stream (padding) without reading it, so that the
// barriers and prefetch pieces scheduled on those positions still happen and the next pass
// finds its first kAhead slices in flight.
template <int F_END, int STREAM_FRAGS = kStreamFrags, class WS>
__device__ __forceinline__ void finish_pass(WS& ws) {
    static_assert(F_END % 2 == 0, "pairs");
    static_for<(STREAM_FRAGS - F_END) / 2>([&](auto I) {
        constexpr int f = F_END + 2 * decltype(I)::value;
        if constexpr (f % kSliceFrags == 0) ws.open_slice();
        ws.template step_piece<f>();
    });
}

// acc[4q..4q+3] of one tile <- bias of channels 32t + 8q + 4h + 0..3
template <int Q>
__device__ __forceinline__ void bias_quad(f32x16& tile, const float* bias_tile_half /* bias_s + off + 32t + 4h */) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_tile_half + 8 * Q);
    tile[4 * Q + 0] = b.x;
    tile[4 * Q + 1] = b.y;
    tile[4 * Q + 2] = b.z;
    tile[4 * Q + 3] = b.w;
}
__device__ __forceinline__ void bias_tile(f32x16& tile, const float* bias_tile_half) {
    bias_quad<0>(tile, bias_tile_half);
    bias_quad<1>(tile, bias_tile_half);
    bias_quad<2>(tile, bias_tile_half);
    bias_quad<3>(tile, bias_tile_half);
}
template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NT], const float* bias_half /* bias_s + off + 4h */) {
    static_for<NT>([&](auto T) { bias_tile(acc[decltype(T)::value], bias_half + 32 * decltype(T)::value); });
}

// ReLU as exactly one VALU instruction the compiler can see: a signed-integer max of the bit
// pattern with 0 (negative floats, -0.0 included, are negative integers).  The float spellings
// (fmaxf, fmed3) are canonicalised by hipcc into v_max_f32 x,x ; v_max_f32 0,x -- twice the issue
// slots -- and an inline-asm v_max_f32 is invisible to the hazard recogniser: where the accumulators
// live in VGPRs (the 8-wave plain-bf16 kernel) it read MFMA results before they were written.
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// ReLU, in place, of registers [R0, R0 + CNT) of a tile (clipped to the 16 a tile has)
template <int R0, int CNT>
__device__ __forceinline__ void relu_regs(f32x16& tile) {
    static_for<CNT>([&](auto R) {
        constexpr int r = R0 + decltype(R)::value;
        if constexpr (r < 16) tile[r] = relu1(tile[r]);
    });
}
template <int NT>
__device__ __forceinline__ void relu_tiles(f32x16 (&t)[NT]) {
    static_for<NT>([&](auto T) { relu_regs<0, 16>(t[decltype(T)::value]); });
}

struct NoSide {
    template <int T, int S, int H>
    __device__ __forceinline__ void operator()(ic<T>, ic<S>, ic<H>) const {}
};

// ---------------------------------------------------------------------------
// Positional encoding gamma_L(v) of a 3-vector, 3 + 6L features in the reference's order
// (helper.py:183-201): [v, sin(2^0 v), cos(2^0 v), ..., sin(2^(L-1) v), cos(2^(L-1) v)].
//
// The reference evaluates sin/cos of fl32(2^b v), and 2^b v is exact, so the true argument is
// known exactly (ocml's sincosf takes its Payne-Hanek path for the upper bands and cost ~10 % of
// the fp32 kernel).  It is evaluated once per feature THE LANE KEEPS: every B-operand slot of the
// input layers holds feature K0 in the lower lane half and K0 + 4 in the upper one, and both lane
// halves carry the same point, so computing all features per lane would do the work twice.
//
//   feature K >= 3:  t = K - 3, band b = t / 6, sc = (t % 6) / 3 (0 sin, 1 cos), axis a = K % 3
//                    value = sin(2 pi (frac(2^b x_a / 2pi) + sc / 4))
//
// The phase frac(x_a / 2pi) is kept as 64-bit fixed point (from fp64, once per axis); band b is
// a funnel shift (v_alignbit), cos is a quarter-turn offset, the reduction to the nearest half
// turn and the sign are integer operations, and ONE odd polynomial on |x| <= pi/2 serves every
// feature (max |diff| to torch.sin/cos on CPU 1.2e-7 on every band, <= 2 ulp at 1.0).  K0 + 4 has
// axis (a + 1) % 3, so the upper half stores its axes rotated by one and both halves index axis
// K0 % 3; band and sc of the two halves enter as per-lane shift / offset operands.
// ---------------------------------------------------------------------------
struct PeLane {  // per-lane constants, set once per kernel
    int h;
    uint32_t off01, off10;  // (sc << 30) + 2^30 per lane half for (sc_lower, sc_upper) = (0, 1) / (1, 0)
    __device__ __forceinline__ void init(int h_) {
        h = h_;
        off01 = h_ ? 0x80000000u : 0x40000000u;
        off10 = h_ ? 0x40000000u : 0x80000000u;
    }
};
struct PeAxes {  // one 3-vector of one point, axes rotated by one in the upper lane half
    uint32_t hi[3], lo[3];
    float raw[3];
    __device__ __forceinline__ void init(const float (&v)[3], int h) {
        uint32_t H[3], Lw[3];
        static_for<3>([&](auto A) {
            constexpr int a = decltype(A)::value;
            const double ph = (double)v[a] * 0.15915494309189535;    // x / 2pi
            const double t = __builtin_amdgcn_fract(ph) * 4294967296.0;  // turns in [0, 1), scaled exactly
            H[a] = (uint32_t)t;
            Lw[a] = (uint32_t)((t - (double)H[a]) * 4294967296.0);
        });
        static_for<3>([&](auto A) {
            constexpr int a = decltype(A)::value, a1 = (a + 1) % 3;
            hi[a] = h ? H[a1] : H[a];
            lo[a] = h ? Lw[a1] : Lw[a];
            raw[a] = h ? v[a1] : v[a];
        });
    }
};
template <int K, int L>
struct PeFeat {
    static constexpr bool raw = K < 3;
    static constexpr bool trig = K >= 3 && K < 3 + 6 * L;
    static constexpr int t = trig ? K - 3 : 0;
    static constexpr int b = t / 6, sc = (t % 6) / 3;
};
template <int B>
__device__ __forceinline__ uint32_t pe_phase(const PeAxes& ax, int a) {
    if constexpr (B == 0) return ax.hi[a];
    else return __builtin_amdgcn_alignbit(ax.hi[a], ax.lo[a], 32 - B);  // (hi << B) | (lo >> (32 - B))
}
// feature K0 (lower lane half) / K0 + 4 (upper lane half) of gamma_L
template <int K0, int L>
__device__ __forceinline__ float pe_slot(const PeAxes& ax, const PeLane& ln) {
    using F0 = PeFeat<K0, L>;
    using F1 = PeFeat<K0 + 4, L>;
    constexpr int a = K0 % 3;
    if constexpr (!F0::trig && !F1::trig) {
        if constexpr (F0::raw) return ln.h ? 0.0f : ax.raw[a];
        else return 0.0f;
    } else {
        constexpr int b0 = F0::trig ? F0::b : F1::b, b1 = F1::trig ? F1::b : F0::b;
        constexpr int sc0 = F0::trig ? F0::sc : F1::sc, sc1 = F1::trig ? F1::sc : F0::sc;
        uint32_t T;
        if constexpr (b0 == b1) T = pe_phase<b0>(ax, a);
        else T = ln.h ? pe_phase<b1>(ax, a) : pe_phase<b0>(ax, a);
        uint32_t u;
        if constexpr (sc0 == sc1) u = T + (((uint32_t)sc0 << 30) + 0x40000000u);
        else u = T + (sc0 == 0 ? ln.off01 : ln.off10);
        // u = phase + half-turn rounding bias: bit 31 = half-turn parity, the rest - 2^30 = residual
        const int rr2 = (int)((u << 1) + 0x80000000u);            // residual * 2, signed, |.| <= 2^31
        const float x = (float)rr2 * 7.3145905512e-10f;           // 2 pi / 2^33 -> |x| <= pi/2
        const float s2 = x * x;
        float pp = -2.391218132e-08f;
        pp = fmaf(pp, s2, 2.752665757e-06f);
        pp = fmaf(pp, s2, -1.984089153e-04f);
        pp = fmaf(pp, s2, 8.333331268e-03f);
        pp = fmaf(pp, s2, -1.666666663e-01f);
        float r = fmaf(x * s2, pp, x);
        r = __uint_as_float(__float_as_uint(r) ^ (u & 0x80000000u));
        if constexpr (!F0::trig) r = ln.h ? r : (F0::raw ? ax.raw[a] : 0.0f);
        if constexpr (!F1::trig) r = ln.h ? 0.0f : r;
        return r;
    }
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

// Raw inputs of one lane's point.  They are loaded one pass ahead (right after a slice opens in
// the middle of the previous pass and "touched" after the next one, where the slice barrier's
// vmcnt(0) has already covered them), so a pass starts with its encoding instead of a global-load
// round trip queued behind the weight pieces issued at the end of the pass before.
struct PointIn {
    float f[10];  // kModeRays: o(3) d(3) viewdir(3) z   kModePts: p(3) viewdir(3)   kModeX: unused
};
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <int MODE, int NW = 4>
__device__ __forceinline__ void load_point(const MlpArgs& a, long tile, int wave, int m, PointIn& in) {
    if constexpr (MODE == kModeX) return;
    long P = tile * (32 * NW) + wave * 32 + m;
    if (P >= a.n_points) P = a.n_points - 1;  // also covers "no next tile": a valid, unused address
    const unsigned ray = (unsigned)P / (unsigned)a.S;  // n_points < 2^31 (checked by the launchers)
    if constexpr (MODE == kModeRays) {
        const float* rr = a.rays + (long)ray * IDN_RAY_FLOATS;
        in.f[0] = rr[0]; in.f[1] = rr[1]; in.f[2] = rr[2];
        in.f[3] = rr[3]; in.f[4] = rr[4]; in.f[5] = rr[5];
        in.f[6] = rr[8]; in.f[7] = rr[9]; in.f[8] = rr[10];
        in.f[9] = a.z[P];
    } else {
        in.f[0] = a.pts[P * 3 + 0]; in.f[1] = a.pts[P * 3 + 1]; in.f[2] = a.pts[P * 3 + 2];
        in.f[3] = a.dirs[(long)ray * 3 + 0]; in.f[4] = a.dirs[(long)ray * 3 + 1]; in.f[5] = a.dirs[(long)ray * 3 + 2];
    }
}
__device__ __forceinline__ void touch_point(PointIn& in) {
#pragma unroll
    for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(in.f[i]));
}
// point and view direction of the lane (pts = rays_o + rays_d * z with product and sum rounded
// separately, audio_exp_nerf.py:332; this code is built with -ffp-contract=off)
template <int MODE>
__device__ __forceinline__ void point_of(const PointIn& in, float (&p)[3], float (&v)[3]) {
    if constexpr (MODE == kModeRays) {
        p[0] = in.f[0] + in.f[3] * in.f[9];
        p[1] = in.f[1] + in.f[4] * in.f[9];
        p[2] = in.f[2] + in.f[5] * in.f[9];
        v[0] = in.f[6]; v[1] = in.f[7]; v[2] = in.f[8];
    } else {
        p[0] = in.f[0]; p[1] = in.f[1]; p[2] = in.f[2];
        v[0] = in.f[3]; v[1] = in.f[4]; v[2] = in.f[5];
    }
}

// This lane's input features: calls use(fpt, fdir) with two providers, fpt(ic<K0>) / fdir(ic<K0>) =
// feature K0 + 4h of gamma_10(point) / gamma_4(view direction) (zero beyond 63 / 27).
template <int MODE, class Use>
__device__ __forceinline__ void input_features(const MlpArgs& a, long Pc, int h, const PeLane& pln, const PointIn& in,
                                               Use&& use) {
    if constexpr (MODE == kModeX) {
        const float* xr = a.x + Pc * (IDN_PTS_CH + IDN_VIEWS_CH) + 4 * h;
        use([&](auto K) { constexpr int k = decltype(K)::value; return (k + 4 * h < IDN_PTS_CH) ? xr[k] : 0.0f; },
            [&](auto K) { constexpr int k = decltype(K)::value; return (k + 4 * h < IDN_VIEWS_CH) ? xr[IDN_PTS_CH + k] : 0.0f; });
    } else {
        float p[3], v[3];
        point_of<MODE>(in, p, v);
        PeAxes axp, axd;
        axp.init(p, h);
        axd.init(v, h);
        use([&](auto K) { return pe_slot<decltype(K)::value, 10>(axp, pln); },
            [&](auto K) { return pe_slot<decltype(K)::value, 4>(axd, pln); });
    }
}

// "Unit off" bits of a tile's 16 pre-activations pushed into mask dword T / 2 (idn_internal.h "ReLU masks"): a unit is off
// where its pre-activation is <= 0 -- exactly +0.0 included, as torch's relu backward (grad * (result > 0)) has it; the plain
// sign bit would pass gradient through a unit that sits at +0.0 (a zero-weight, zero-bias layer).  x <= 0 <=> max_i32(bits, 0)
// == 0 <=> the sign bit of max_i32(bits, 0) - 1; one v_alignbit per value pushes it: (mk << 1) | bit.
__device__ __forceinline__ uint32_t off_bit_source(float x) {
    const int b = __builtin_bit_cast(int, x);
    return (uint32_t)((b > 0 ? b : 0) - 1);
}
template <int T>
__device__ __forceinline__ void collect_signs(const f32x16& tile, uint32_t* mk) {
    static_for<16>([&](auto R) {
        mk[T >> 1] = __builtin_amdgcn_alignbit(mk[T >> 1], off_bit_source(tile[decltype(R)::value]), 31);
    });
}

// buffer descriptor over the 32 rows (LD floats each) a wave writes of a row-major matrix
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(float* first_row, int ld) {
    return __builtin_amdgcn_make_buffer_rsrc(first_row, 0, 32 * ld * 4, 0x00020000);
}

constexpr int kMlpLds = kRingFrags * kFragBytes + kBiasFloats * 4;
constexpr int kStagePitch = 33;                       // 32x32 transpose tile, conflict-free
constexpr int kStageFloats = 32 * kStagePitch;
constexpr int kMlpLdsTrain = kMlpLds + 4 * kStageFloats * 4;


}  // namespace idn
