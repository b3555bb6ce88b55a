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
// The ring holds kRingSlots slices (idn_internal.h).  A slice is fetched by "pieces" (one
// global_load_lds_dwordx4 per wave each = 4 KiB per piece); while slice s is consumed, the
// pieces of slice s + kRingSlots-1 are issued, one per fragment-pair step, into the slot slice
// s-1 just vacated.  A slice is opened by a COUNTED wait + raw barrier: s_waitcnt vmcnt(N)
// leaves the N pieces of the younger slices in flight (N = 0 for the two-slot ring).  Other
// vector-memory operations of the wave only make the counted wait stricter.
// ---------------------------------------------------------------------------
constexpr int kPieces = kSliceFrags / 4;                    // pieces per slice
#ifndef IDN_PIECES_PER_STEP
#define IDN_PIECES_PER_STEP 1
#endif
constexpr int kPiecesPerStep = IDN_PIECES_PER_STEP;         // pieces issued per fragment-pair step
static_assert(kPieces % kPiecesPerStep == 0, "a slice is a whole number of steps");
constexpr int kAhead = kRingSlots - 1;                      // slices in flight ahead of the consumer
constexpr int kVmcntOpen = (kAhead - 1) * kPieces;          // younger pieces allowed in flight at a barrier
static_assert(kVmcntOpen == 0 || kVmcntOpen == 8 || kVmcntOpen == 16, "add the s_waitcnt literal below");

struct WStream {
    Diag* dg;
    // Pieces are buffer_load_dwordx4 ... lds: descriptor (4 SGPRs) + one constant VGPR (lane * 16) +
    // a scalar stream offset.  Measured beside an MFMA chain (tools/glds_ubench.hip) a piece in
    // this form costs the issuing wave ~1.5 cycles against ~8 for global_load_lds with per-lane
    // 64-bit pointers, and the clock holds ~10 % higher.  The descriptor's num_records bounds
    // every piece to the stream (an out-of-range piece reads zeros instead of faulting).
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;      // this lane's 16-byte column inside a piece
    uint32_t soff;      // byte offset of the slice currently being fetched (wave-uniform)
    int next_slice;
    int num_slices;     // slices in this stream (kNumSlices, or kPlainNumSlices for the plain-bf16 stream)
    char* ring_wave;    // ring + wave * 1 KiB (wave-uniform LDS destination base)
#ifdef IDN_DIAG_NOSTREAM
    bool pass_done = true;  // set false to drop all prefetch pieces after the prologue
#endif

    __device__ __forceinline__ void init(const float* stream, int slices, char* ring, int tid, int wave) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, slices * kSliceBytes, 0x00020000);  // raw, untyped
        voff = tid * 16;
        soff = 0;
        next_slice = 0;
        num_slices = slices;
        ring_wave = ring + wave * kFragBytes;
        prologue();
    }
    __device__ __forceinline__ void advance() {
        soff += kSliceBytes;
        if (++next_slice == num_slices) {
            next_slice = 0;
            soff = 0;
        }
    }
    template <int SLOT, int J>
    __device__ __forceinline__ void issue_piece() {
#ifdef IDN_DIAG_NOSTREAM  // timing-only experiment: what does the weight stream cost? (outputs are garbage)
        if (pass_done)
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(ring_wave + SLOT * kSliceBytes + J * 4096), 16, voff,
                                                 soff + J * 4096, 0, 0);
        if constexpr (J == kPieces - 1) advance();
    }
    template <int SLOT, int J0>
    __device__ __forceinline__ void issue_rest() {
        static_for<kPieces - J0>([&](auto I) { issue_piece<SLOT, J0 + decltype(I)::value>(); });
    }
    // kernel start: slices 0 .. kAhead-1 into slots 0 .. kAhead-1
    __device__ __forceinline__ void prologue() {
#ifdef IDN_STAGGER  // experiment: spread the workgroups of an XCD over one slice period
        for (int i = (blockIdx.x >> 3) & 31; i > 0; --i) __builtin_amdgcn_s_sleep(IDN_STAGGER);
#endif
        static_for<kAhead>([&](auto S_) { issue_rest<decltype(S_)::value, 0>(); });
#ifdef IDN_DIAG_NOSTREAM
        pass_done = false;
#endif
    }
    // Open the next slice: this wave's pieces of it have landed (all but the 16 youngest
    // vector-memory operations are complete); after the barrier every wave's have, and every
    // wave is done reading the slot the next pieces will overwrite.
    __device__ __forceinline__ void open_slice() {
        DIAG_BEGIN(*dg);
#ifdef IDN_DIAG_NOWAIT  // timing-only experiment: pieces are issued but never waited for (outputs are garbage)
        if constexpr (false) {}
#else
        if constexpr (kVmcntOpen == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        else if constexpr (kVmcntOpen == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DIAG_END(*dg, kDgBarrier);
    }
    // the piece (if any) to issue at the pair-step that consumes fragment F
    template <int F>
    __device__ __forceinline__ void step_piece() {
        constexpr int jpos = (F % kSliceFrags) / 2;
        constexpr int slot = (F / kSliceFrags + kAhead) % kRingSlots;
        if constexpr (jpos * kPiecesPerStep < kPieces)
            static_for<kPiecesPerStep>([&](auto I) { issue_piece<slot, jpos * kPiecesPerStep + decltype(I)::value>(); });
    }
};

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
    f32x4 pref0, pref1;     // fragment pair issued ahead of its consumer (layer / slice start)

    template <int F>
    __device__ __forceinline__ f32x4 issue() const {
        constexpr int fr = F % kRingFrags;
        f32x4 v;
        if constexpr (fr < 64)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr0), "n"(fr * kFragBytes) : "memory");
        else
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr1), "n"((fr - 64) * kFragBytes) : "memory");
        return v;
    }
    // all but the newest `Newer` LDS reads of this wave have completed => v0, v1 are valid
    template <int Newer>
    static __device__ __forceinline__ void retire(f32x4& v0, f32x4& v1) {
        if constexpr (Newer == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v0), "+v"(v1)::"memory");
    }
};

// End of a pass: walk the unused tail of the stream (padding) without reading it, so that the
// barriers and prefetch pieces scheduled on those positions still happen and the next pass
// finds its first kAhead slices in flight.
template <int F_END, int STREAM_FRAGS = kStreamFrags>
__device__ __forceinline__ void finish_pass(WStream& ws) {
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

// ReLU as exactly one VALU instruction.  Written as asm because every builtin spelling (fmaxf,
// fmed3) is canonicalised by hipcc into v_max_f32 x,x ; v_max_f32 0,x -- twice the issue slots,
// in the part of the kernel where VALU issue is what the MFMA chain competes with.
__device__ __forceinline__ float relu1(float x) {
#ifdef IDN_RELU_BUILTIN  // A/B arm
    return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff());
#else
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
#endif
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

// sin / cos of 2*pi*r for |r| <= 1/8 (Taylor in r; the first dropped terms are < 2e-9).
__device__ __forceinline__ void sincos_2pi_small(float r, float& sn, float& cs) {
    const float s = r * r;
    float ps = 4.2058693945e+01f;                 //  (2pi)^9 / 9!
    ps = fmaf(ps, s, -7.6705859753e+01f);         // -(2pi)^7 / 7!
    ps = fmaf(ps, s, 8.1605249276e+01f);          //  (2pi)^5 / 5!
    ps = fmaf(ps, s, -4.1341702240e+01f);         // -(2pi)^3 / 3!
    ps = fmaf(ps, s, 6.2831853072e+00f);          //   2pi
    sn = ps * r;
    float pc = -2.6426256783e+01f;                // -(2pi)^10 / 10!
    pc = fmaf(pc, s, 6.0244641371e+01f);          //  (2pi)^8 / 8!
    pc = fmaf(pc, s, -8.5456817206e+01f);         // -(2pi)^6 / 6!
    pc = fmaf(pc, s, 6.4939394023e+01f);          //  (2pi)^4 / 4!
    pc = fmaf(pc, s, -1.9739208802e+01f);         // -(2pi)^2 / 2!
    cs = fmaf(pc, s, 1.0f);
}

// gamma_L(v) for a 3-vector, as 3 + 6L features in the reference's order
// (helper.py:183-201): [v, sin(2^0 v), cos(2^0 v), ..., sin(2^(L-1) v), cos(2^(L-1) v)],
// zero padded to NF.
//
// The reference evaluates sin/cos of fl32(2^b v), and 2^b v is exact, so the true
// argument is known exactly: reduce the phase p = v / 2pi once per axis in fp64, double it
// per band (exact), split off the quadrant (exact) and evaluate a short polynomial on
// |r| <= 1/8 of a turn.  Measured against torch.sin/cos on CPU: max |diff| 1.2e-7 on every
// band (<= 2 ulp at 1.0), with no data-dependent branch (ocml's sincosf takes its
// Payne-Hanek path for the upper bands and cost ~10 % of the kernel).
template <int L, int NF>
__device__ __forceinline__ void encode(const float (&v)[3], float (&feat)[NF]) {
    static_assert(NF >= 3 + 6 * L, "feature buffer too small");
    static_for<NF>([&](auto K) { feat[decltype(K)::value] = 0.0f; });
    feat[0] = v[0];
    feat[1] = v[1];
    feat[2] = v[2];
    static_for<3>([&](auto A) {
        constexpr int a = decltype(A)::value;
        double p = (double)v[a] * 0.15915494309189535;  // 1 / 2pi
        static_for<L>([&](auto B) {
            constexpr int b = decltype(B)::value;
            p = p - rint(p);                       // [-1/2, 1/2] turns, exact
            const double q = rint(p * 4.0);        // nearest quarter turn
            const float r = (float)(p - q * 0.25); // [-1/8, 1/8], exact before the conversion
            const int qi = (int)q & 3;
            float sn, cs;
            sincos_2pi_small(r, sn, cs);
            const float s_out = (qi & 1) ? cs : sn;
            const float c_out = (qi & 1) ? sn : cs;
            feat[3 + 6 * b + a] = (qi == 2 || qi == 3) ? -s_out : s_out;
            feat[3 + 6 * b + 3 + a] = (qi == 1 || qi == 2) ? -c_out : c_out;
            p = p + p;
        });
    });
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

constexpr int kMlpLds = kRingFrags * kFragBytes + kBiasFloats * 4;
constexpr int kStagePitch = 33;                       // 32x32 transpose tile, conflict-free
constexpr int kStageFloats = 32 * kStagePitch;
constexpr int kMlpLdsTrain = kMlpLds + 4 * kStageFloats * 4;


}  // namespace idn
