// Backward of one render pass (coarse or fine) for the training step
// (NeRFs/HeadNeRF/train/audio_exp_nerf.py:534-552: loss.backward() through raw2outputs,
// FaceNeRF and the per-frame conditioning).  gfx950, fp32 MFMA.
//
//   composite_bwd     d(rgb_map, rgb_fg, last_weight, acc) -> d raw           (one wave per ray)
//   delta chain       all delta_l = (delta_{l+1} . W_{l+1}) (.) [a_l > 0] in one fused kernel (mlp_f32_bwd.hip)
//   gemm_tn           dW_l = delta_l^T . a_{l-1}, contraction over points, split over workgroups
//   reduce_batch      sums the per-split dW blocks of all layers into the gradient tensors (one launch per pass)
//   (db_l = sum_p delta_l comes out of gemm_tn's pass over delta_l)
//   fold_bwd          conditioning columns of W0 / W5 / Wv0 and d aud, d latent
//
// Activations come from the training variant of the MLP kernel as row-major matrices
// (idn_internal.h, "activation slab").  Everything is deterministic: no float atomics.
#include "idn_internal.h"
#include <cstdlib>
#include <type_traits>

namespace idn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int d_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------
// TN GEMM: part[split][n][k] = sum_{p in split} A[p][n0+n] * B[p][k0+k]
// Block = (64*NTW) x (64*KTW) outputs, waves 2 x 2, contraction chunk 32 points.
// ---------------------------------------------------------------------------
struct TNArgs {
    const float* A; int lda;   // delta  [P, >= N]
    const float* B; int ldb;   // acts   [P, >= K]
    float* part;               // [splits][N][K]
    int N, K;
    long P;                    // rows (multiple of 32)
    int chunks_per_split;      // kTnRows-row chunks per split
    float* cpart;              // optional [splits][N]: column sums of A (the bias gradient), from the k-block-0 workgroups
    // gemm_tn_x6_kernel only: B as TWO 128-column matrices (row pitch ldb each), columns 0..127 at B + b_off0 bytes and
    // columns 128..255 at B + b_off1 bytes -- two 128 x 128 products as the diagonal blocks of one 256 x 256 launch
    int b_split, b_off0, b_off1;
    // (skipping the MFMAs of the unwanted tiles -- the off-diagonal blocks of a paired launch, rows 129..255 of views_linears.0 +
    //  alpha_linear -- behind wave-uniform branches was tried: the accumulators then flow through phis, hipcc copies registers whose
    //  asm loads are in flight (556 sites in tools/audit_asm_loads.py, results no longer reproducible) and the kernel ran 1.6x slower)
};

// The two 32-row chunk tiles are double-buffered in LDS and filled by LDS-DMA (a chunk row is
// contiguous in global memory and in the tile, so one wave instruction moves 1 KiB of it): the
// loads of chunk c+1 are in flight while chunk c is multiplied, one barrier per chunk.  (Single
// buffered, every chunk paid its global-load latency: 70 % of the fp32 MFMA peak.)
//
// Operand reads.  MFMA row i of a wave's tile x is output channel NTW * i + x (not 32 x + i): a lane's NTW
// A values of one point are then CONTIGUOUS in the row-major LDS tile and come with one ds_read_b128
// (b64 / b32) instead of NTW strided ds_read_b32; the same for B.  The reads of point-pair s + 1 are
// issued before the MFMAs of pair s from inline asm and retired by a counted wait tied to the
// destination registers (hipcc issues such reads right before their use and waits lgkmcnt(0): an exposed
// LDS round trip every 8 MFMAs, 12 % of the wave cycles parked, measured with SQ_WAIT_ANY).
template <int N>
struct LdsVec;
template <>
struct LdsVec<4> {
    f32x4 v;
    template <int OFF>
    __device__ __forceinline__ void issue(uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory"); }
    __device__ __forceinline__ float get(int j) const { return v[j]; }
};
template <>
struct LdsVec<2> {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v;
    template <int OFF>
    __device__ __forceinline__ void issue(uint32_t addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory"); }
    __device__ __forceinline__ float get(int j) const { return v[j]; }
};
template <>
struct LdsVec<1> {
    float v;
    template <int OFF>
    __device__ __forceinline__ void issue(uint32_t addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory"); }
    __device__ __forceinline__ float get(int) const { return v; }
};
template <int OFF>
__device__ __forceinline__ void lds_read_f32(float& dst, uint32_t addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// use of an asm-read value: not before the counted wait that precedes this call in program order
__device__ __forceinline__ float landed(float& v) {
    asm volatile("" : "+v"(v));
    return v;
}
// all but the newest `NEWER` LDS reads of this wave have completed => a, b are valid
template <int NEWER, class VA, class VB>
__device__ __forceinline__ void lds_retire(VA& a, VB& b) {
    if constexpr (NEWER == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v), "+v"(b.v)::"memory");
    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a.v), "+v"(b.v)::"memory");
}
template <int N>
using ic_ = std::integral_constant<int, N>;
template <int N, class F>
__device__ __forceinline__ void tn_static_for(F&& f) {
    if constexpr (N > 0) {
        tn_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Chunks of kTnRows points; kTnBufs LDS buffers: while chunk c is multiplied, the pieces of chunk c + 2 are
// issued ONE PER POINT-PAIR STEP (a burst of 16 LDS-DMA instructions at the top of a chunk held the wave's
// MFMA issue for ~10 % of the chunk), as `buffer_load ... lds` with a per-lane constant offset and a scalar
// row offset (half the issue cost of the per-lane-pointer form), and they have a whole chunk to land: the
// GEMM reads 2 KiB per point and layer for 131 kFLOP, i.e. it needs 2.5 TB/s of HBM at the MFMA peak.
constexpr int kTnRows = 16, kTnBufs = 3;
#ifndef IDN_DELTA_X6
#define IDN_DELTA_X6 1  // the backward delta chain as six bf16 piece products per fp32 product (delta_chain_x6_kernel); 0: fp32 MFMA
#endif
#ifndef IDN_DW_X6
#define IDN_DW_X6 1    // 256 x 256 dW GEMMs as six bf16 piece products (gemm_tn_x6_kernel); 0: the fp32-MFMA kernel
#endif

#ifdef IDN_DIAG   // diagnostic build only: where a <4,4> block spends its cycles (tools/diag_tn.py)
__device__ unsigned long long g_tn_diag[8];   // total, wait (vmcnt + barrier), loop, epilogue, blocks, chunks
#define TN_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TN_STAMP(v)
#endif

// NB = LDS buffers of the chunk ring: the pieces of chunk c + NB - 1 are issued while chunk c is multiplied.  The 256 x 256
// shape (32 KiB per chunk, 16 MFMAs per point pair and wave) runs NB = 3; the narrow shapes do a quarter of the arithmetic per
// byte (a 256 x 64 chunk is 20 KiB for 4 MFMAs per point pair: 1 us of matrix time, less than a loaded HBM round trip) and
// run a deeper ring, so that several chunks per workgroup are in flight.
// R = points per chunk (a multiple of 16).  A chunk costs ~1 000 cycles besides its MFMAs (the barrier and its skew, the first LDS
// reads behind it with nothing to overlap, the drain of the prefetched reads at its end: in-kernel stamps, tools/diag_tn.py:
// 3 007 cycles per 16-point chunk of the 256 x 64 shape against 2 048 of MFMAs), so the shapes with few MFMAs per point take
// larger chunks.
template <int NTW, int KTW, int NB = kTnBufs, int R = kTnRows>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TNArgs g) {
    static_assert(R % 16 == 0, "chunks are multiples of 16 points");
    constexpr int NPA = NTW * R / 16, NPB = KTW * R / 16;   // 1-KiB pieces of the A / B tile per wave and chunk
    static_assert(NB >= 3 && (NB - 2) * (NPA + NPB) <= 63, "vmcnt is a 6-bit field");
    constexpr int BN = 64 * NTW, BK = 64 * KTW;
    constexpr int kTileFloats = R * (BN + BK);       // one chunk: A tile then B tile
    constexpr int kSteps = R / 2;                    // point-pairs per chunk
    constexpr int NP = NPA + NPB;                          // 1-KiB pieces per wave and chunk (<= kSteps)
    static_assert(NP <= kSteps, "at most one piece per step");
    extern __shared__ __attribute__((aligned(16))) float tn_smem[];  // NB * kTileFloats
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    const int n0 = blockIdx.x * BN, k0 = blockIdx.y * BK;
    const int split = blockIdx.z;
    f32x16 acc[NTW][KTW];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < KTW; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    unsigned long long dg_t0 = 0, dg_a = 0, dg_b = 0, dg_wait = 0, dg_loop = 0, dg_t1 = 0, dg_t2 = 0;
    (void)dg_t0; (void)dg_a; (void)dg_b; (void)dg_wait; (void)dg_loop; (void)dg_t1; (void)dg_t2;
    TN_STAMP(dg_t0);
    const long c_begin = (long)split * g.chunks_per_split;
    long c_end = c_begin + g.chunks_per_split;
    const long c_total = g.P / R;
    if (c_end > c_total) c_end = c_total;
    const bool colsum_block = g.cpart != nullptr && blockIdx.y == 0;   // block-uniform
    const bool do_colsum = colsum_block && tid < BN;
    const int ccol = tid < BN ? tid : 0;                                  // threads beyond the tile re-read column 0 (unused)
    float csum = 0.0f;

    // A piece = 256 consecutive floats of a tile = 256 / BN rows of it.  Piece j (0 .. NTW-1) of wave w is
    // tile piece NTW * w + j: rows (NTW * w + j) * (256 / BN) ..; lane l moves float4 l of the piece.
    constexpr int kRowsPerPieceA = 256 / BN > 0 ? 256 / BN : 1, kRowsPerPieceB = 256 / BK > 0 ? 256 / BK : 1;
    // descriptors based at this block's first row: the 32-bit piece offsets then span one split (tens of MB), not the
    // whole matrix (which passes 4 GB from 4 M points on)
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + n0 + c_begin * R * (long)g.lda), 0, 0xfffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B + k0 + c_begin * R * (long)g.ldb), 0, 0xfffffffc, 0x00020000);
    const uint32_t voffA = ((lane / (BN / 4)) * g.lda + (lane % (BN / 4)) * 4) * 4;
    const uint32_t voffB = ((lane / (BK / 4)) * g.ldb + (lane % (BK / 4)) * 4) * 4;
    const uint32_t rowA = g.lda * 4, rowB = g.ldb * 4;    // bytes per matrix row
    // piece ja (0 .. NTW-1) of this wave's share of the A tile / piece jb (0 .. KTW-1) of the B tile, for chunk c
    // into buffer `buf`: scalar arithmetic only (a piece that had to choose between the two matrices cost a
    // tree of scalar branches per point pair, ~10 % of the loop)
    auto piece_a = [&](long c, int buf, int ja) {
#ifdef IDN_TN_TIMING_NO_PIECES   // timing-only experiment (wrong results): the GEMM without its HBM traffic
        if (c > c_begin + 1) return;
#endif
        const int pc = NPA * w + ja;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(tn_smem + buf * kTileFloats + pc * 256), 16, voffA,
                                                 (uint32_t)(((c - c_begin) * R + pc * kRowsPerPieceA) * rowA), 0, 0);
    };
    auto piece_b = [&](long c, int buf, int jb) {
#ifdef IDN_TN_TIMING_NO_PIECES
        if (c > c_begin + 1) return;
#endif
        const int pc = NPB * w + jb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (__attribute__((address_space(3))) void*)(tn_smem + buf * kTileFloats + R * BN + pc * 256), 16, voffB,
                                                 (uint32_t)(((c - c_begin) * R + pc * kRowsPerPieceB) * rowB), 0, 0);
    };
    auto piece = [&](long c, int buf, int j) {   // prologue order: A pieces, then B pieces
        if (j < NPA) piece_a(c, buf, j);
        else if (j < NP) piece_b(c, buf, j - NPA);
    };
    // LDS byte addresses of this lane's operands of point-pair 0 in buffer 0: row hh, columns NTW * (32 wr + i) ..
    const uint32_t smem0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)tn_smem;
    const uint32_t a_addr0 = smem0 + (hh * BN + NTW * (32 * wr + i)) * 4;
    const uint32_t b_addr0 = smem0 + (R * BN + hh * BK + KTW * (32 * wc + i)) * 4;
    // every chunk's NP pieces are issued even past the end of the split (clamped to its last chunk: re-read, never used), so
    // that exactly (NB - 2) * NP younger vector-memory operations are in flight at every chunk's wait
    for (int ahead = 0; ahead < NB - 1; ++ahead)
        for (int j = 0; j < NP; ++j) piece(c_begin + ahead < c_end ? c_begin + ahead : c_end - 1, ahead, j);
    int buf = 0;
    for (long c = c_begin; c < c_end; ++c) {
        TN_STAMP(dg_a);
        // this wave's pieces of chunk c have landed (those of chunks c + 1 .. c + NB - 2, issued later, may still be in flight)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * NP) : "memory");
        // everyone's have; everyone is done with chunk c - 1, whose buffer chunk c + 2 now takes.  A raw barrier:
        // __syncthreads() would add its own vmcnt(0) and wait for the pieces of chunk c + 1 as well
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TN_STAMP(dg_b);
#ifdef IDN_DIAG
        dg_wait += dg_b - dg_a;
#endif
        const int buf2 = buf >= 1 ? buf - 1 : NB - 1;   // (buf + NB - 1) % NB: the buffer chunk c - 1 has just left
        const long cnext = c + NB - 1 < c_end ? c + NB - 1 : c_end - 1;   // clamped: the re-read of the last chunk is never used
        constexpr bool more = true;
        // The delta tile is in LDS anyway: its column sums are the bias gradient.  The R reads of column
        // `tid` go out first (inline asm, like the operand reads) and are added after the first counted wait of
        // the loop below, which covers them (LDS returns in order): read by plain loads they parked every wave
        // for two LDS round trips per chunk, 5 % of it.
        // (In batches of 16 rows: batch k + 1 is issued at the end of loop iteration k -- in front of that iteration's operand
        // reads, whose counted wait in iteration k + 1 then covers it -- into the registers batch k has just been added from.)
        float cs[16];
        auto cs_issue = [&](int batch) {
            const uint32_t caddr = smem0 + buf * (kTileFloats * 4) + ccol * 4 + batch * (16 * BN * 4);
            tn_static_for<16>([&](auto Row) {
                lds_read_f32<decltype(Row)::value * BN * 4>(cs[decltype(Row)::value], caddr);
            });
        };
        static_assert(R / 16 <= kSteps / 4, "one batch of column reads per iteration of the first half loop");
        if (colsum_block) cs_issue(0);
        // kSteps point-pairs, two per loop iteration (one per register buffer); the loop is kept rolled: fully
        // unrolled, hipcc shuffles the 256 accumulator registers between steps (~500 v_accvgpr_mov per chunk)
        LdsVec<NTW> a0v, a1v;
        LdsVec<KTW> b0v, b1v;
        constexpr int kStepA = 2 * BN * 4, kStepB = 2 * BK * 4;   // bytes from one point-pair to the next
        uint32_t pa = a_addr0 + buf * (kTileFloats * 4), pb = b_addr0 + buf * (kTileFloats * 4);
        a0v.template issue<0>(pa);
        b0v.template issue<0>(pb);
        // first half of the chunk: the A pieces of chunk c + 2 (one per point pair), second half: its B pieces
        static_assert(NPA <= kSteps / 2 && NPB <= kSteps / 2, "pieces of one matrix fit one half of a chunk");
        constexpr int M = NTW * KTW;        // MFMAs per point pair
        if constexpr (M >= 4) {
            // The fp32 MFMA issues every 64 cycles and everything else issues IN ORDER between two of them: left to hipcc, a
            // pair's MFMAs go out back to back and its ~20 other instructions (operand reads, the LDS-DMA piece with its scalar
            // address arithmetic, pointer updates, the loop branch) follow in one run -- longer than the 60 cycles the last MFMA
            // leaves free, so the matrix pipe idled ~120 cycles per point pair (in-kernel stamps: 376 cycles per pair of the
            // 256 x 64 shape against 256; the same ~120 on the 1 024 of the 256 x 256 shape).  The other work is therefore cut into
            // three slots placed behind the first three MFMAs of a pair, with scheduling fences; the pair's operands are read in
            // the shadow of the pair before (three MFMAs = 190 cycles ahead of their first use).
            auto step = [&](LdsVec<NTW>& ca, LdsVec<KTW>& cb, auto&& side0, auto&& side1, auto&& side2) {
                lds_retire<0>(ca, cb);
                tn_static_for<M>([&](auto M_) {
                    constexpr int m = decltype(M_)::value, x = m / KTW, y = m % KTW;
                    acc[x][y] = mfma32(ca.get(x), cb.get(y), acc[x][y]);
                    if constexpr (m < 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (m == 0) side0();
                        if constexpr (m == 1) side1();
                        if constexpr (m == 2) side2();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            };
#pragma unroll 1
            for (int it = 0; it < kSteps / 4; ++it) {
                step(a0v, b0v,
                     [&]() { a1v.template issue<kStepA>(pa); b1v.template issue<kStepB>(pb); },
                     [&]() { if (more && 2 * it < NPA) piece_a(cnext, buf2, 2 * it); },
                     [&]() {
                         if (colsum_block && it < R / 16) {   // batch `it` of the column reads is older than a0v / b0v: retired with them
                             tn_static_for<16>([&](auto Row) { csum += landed(cs[decltype(Row)::value]); });
                         }
                         pa += 2 * kStepA;
                         pb += 2 * kStepB;
                     });
                step(a1v, b1v,
                     [&]() { a0v.template issue<0>(pa); b0v.template issue<0>(pb); },
                     [&]() { if (more && 2 * it + 1 < NPA) piece_a(cnext, buf2, 2 * it + 1); },
                     [&]() { if (colsum_block && it + 1 < R / 16) cs_issue(it + 1); });
            }
#pragma unroll 1
            for (int it = 0; it < kSteps / 4; ++it) {
                const bool last = it == kSteps / 4 - 1;
                step(a0v, b0v,
                     [&]() { a1v.template issue<kStepA>(pa); b1v.template issue<kStepB>(pb); },
                     [&]() { if (more && 2 * it < NPB) piece_b(cnext, buf2, 2 * it); },
                     [&]() {
                         // the pair after next; past the last pair the read is repeated on the current rows (never used):
                         // one loop shape for all iterations keeps the accumulators where they are
                         pa = last ? pa : pa + 2 * kStepA;
                         pb = last ? pb : pb + 2 * kStepB;
                     });
                step(a1v, b1v,
                     [&]() { a0v.template issue<0>(pa); b0v.template issue<0>(pb); },
                     [&]() { if (more && 2 * it + 1 < NPB) piece_b(cnext, buf2, 2 * it + 1); },
                     [&]() {});
            }
        } else {
            // Two MFMAs per point pair (the 128 x 64 and 64 x 128 shapes): a pair is too short to hide anything behind, so a step
            // is TWO pairs -- four MFMAs, the reads of the next two pairs behind the first two of them, the piece behind the third,
            // pointer / column-sum work behind the fourth -- on four register sets.
            static_assert(M == 2 && kSteps % 8 == 0, "the two-pair schedule");
            LdsVec<NTW> a2v, a3v;
            LdsVec<KTW> b2v, b3v;
            a1v.template issue<kStepA>(pa);
            b1v.template issue<kStepB>(pb);
            auto quad = [&](LdsVec<NTW>& c0a, LdsVec<KTW>& c0b, LdsVec<NTW>& c1a, LdsVec<KTW>& c1b, LdsVec<NTW>& n0a, LdsVec<KTW>& n0b,
                            LdsVec<NTW>& n1a, LdsVec<KTW>& n1b, bool last, auto&& side_c, auto&& side_d) {
                lds_retire<0>(c0a, c0b);
                lds_retire<0>(c1a, c1b);
                // the two pairs after these; past the chunk's last pair the reads are repeated on the current rows (never used)
                const uint32_t na = last ? pa : pa + 2 * kStepA, nb = last ? pb : pb + 2 * kStepB;
                auto mf = [&](LdsVec<NTW>& ca, LdsVec<KTW>& cb, auto M_) {
                    constexpr int m = decltype(M_)::value, x = m / KTW, y = m % KTW;
                    acc[x][y] = mfma32(ca.get(x), cb.get(y), acc[x][y]);
                    __builtin_amdgcn_sched_barrier(0);
                };
                mf(c0a, c0b, ic_<0>{});
                n0a.template issue<0>(na);
                n0b.template issue<0>(nb);
                __builtin_amdgcn_sched_barrier(0);
                mf(c0a, c0b, ic_<1>{});
                n1a.template issue<kStepA>(na);
                n1b.template issue<kStepB>(nb);
                __builtin_amdgcn_sched_barrier(0);
                mf(c1a, c1b, ic_<0>{});
                side_c();
                __builtin_amdgcn_sched_barrier(0);
                mf(c1a, c1b, ic_<1>{});
                side_d();
                pa = na;
                pb = nb;
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll 1
            for (int it = 0; it < kSteps / 8; ++it) {
                quad(a0v, b0v, a1v, b1v, a2v, b2v, a3v, b3v, false,
                     [&]() { if (more && 2 * it < NPA) piece_a(cnext, buf2, 2 * it); },
                     [&]() {
                         if (colsum_block && it < R / 16) {   // batch `it` of the column reads is older than these operand reads: retired with them
                             tn_static_for<16>([&](auto Row) { csum += landed(cs[decltype(Row)::value]); });
                         }
                     });
                quad(a2v, b2v, a3v, b3v, a0v, b0v, a1v, b1v, false,
                     [&]() { if (more && 2 * it + 1 < NPA) piece_a(cnext, buf2, 2 * it + 1); },
                     [&]() { if (colsum_block && it + 1 < R / 16) cs_issue(it + 1); });
            }
#pragma unroll 1
            for (int it = 0; it < kSteps / 8; ++it) {
                quad(a0v, b0v, a1v, b1v, a2v, b2v, a3v, b3v, false,
                     [&]() { if (more && 2 * it < NPB) piece_b(cnext, buf2, 2 * it); }, [&]() {});
                quad(a2v, b2v, a3v, b3v, a0v, b0v, a1v, b1v, it == kSteps / 8 - 1,
                     [&]() { if (more && 2 * it + 1 < NPB) piece_b(cnext, buf2, 2 * it + 1); }, [&]() {});
            }
            lds_retire<0>(a1v, b1v);
        }
        lds_retire<0>(a0v, b0v);   // drain the repeated read before these registers are reused
        buf = buf + 1 == NB ? 0 : buf + 1;
        TN_STAMP(dg_a);
#ifdef IDN_DIAG
        dg_loop += dg_a - dg_b;
#endif
    }
    TN_STAMP(dg_t1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-reads of the last chunk: nothing may land in LDS after this workgroup has left
    if (do_colsum) g.cpart[(long)split * g.N + n0 + tid] = csum;
    // this lane holds, for tile (x, y) register r: output (n0 + NTW (32 wr + d_row(r, hh)) + x, k0 + KTW (32 wc + i) + y):
    // the KTW values of one (x, r) are contiguous in a row of the partial block
    float* out = g.part + (long)split * g.N * g.K;
#pragma unroll
    for (int x = 0; x < NTW; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + NTW * (32 * wr + d_row(r, hh)) + x;
            float* dst = out + (long)n * g.K + k0 + KTW * (32 * wc + i);
            if constexpr (KTW == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{acc[x][0][r], acc[x][1][r], acc[x][2][r], acc[x][3][r]};
            } else {
#pragma unroll
                for (int y = 0; y < KTW; ++y) dst[y] = acc[x][y][r];
            }
        }
#ifdef IDN_DIAG
    __builtin_amdgcn_s_waitcnt(0);
    TN_STAMP(dg_t2);
#ifndef IDN_DIAG_KTW
#define IDN_DIAG_KTW 4      // which 4-row-tile shape the stamps are collected for: <4,4> (default) or <4,1> (-DIDN_DIAG_KTW=1)
#endif
    if (NTW == 4 && KTW == IDN_DIAG_KTW && tid == 0) {
        atomicAdd(&g_tn_diag[0], dg_t2 - dg_t0);
        atomicAdd(&g_tn_diag[1], dg_wait);
        atomicAdd(&g_tn_diag[2], dg_loop);
        atomicAdd(&g_tn_diag[3], dg_t2 - dg_t1);
        atomicAdd(&g_tn_diag[4], 1ull);
        atomicAdd(&g_tn_diag[5], (unsigned long long)(c_end - c_begin));
    }
#endif
}

// ---------------------------------------------------------------------------
// The 256 x 256 dW GEMMs on the bf16 matrix pipe: every fp32 operand is the exact sum of three bf16 pieces
// (x = p1 + p2 + p3, each the round-to-nearest bf16 of what the pieces before it left: 3 x 8 significand bits
// and bf16 has fp32's exponent range, so nothing is scaled and nothing can overflow that fp32 holds), and a
// product keeps the six piece products down to 2^-16 of it,
//      a.b ~ a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1      (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: the rounding of an fp32 fma chain (2^-24 per step), at
// 16 / 6 of the fp32 MFMA rate.  A block owns the whole 256 x 256 output of one split of the points.
//
// Data path per 16-point chunk: thread t loads column t of the delta tile and of the activation tile (16 dwords
// each, a row of the tile per wave instruction), splits them and stores the pieces FRAGMENT-READY in LDS --
// [matrix][32-channel tile][piece][lane (i, hh)][8 bf16 = points 8 hh .. 8 hh + 7 of channel i]: the thread's own
// 16 bytes per piece and lane half, conflict free -- while the chunk before is multiplied; the registers then take
// the chunk after next straight away, so a load has a whole chunk to land.  Two LDS buffers of 48 KiB, one
// counted wait + raw barrier per chunk.  The bias gradient (column sums of delta) is added up by the thread
// that holds the column anyway.
// ---------------------------------------------------------------------------
typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned tn_u32x4 __attribute__((ext_vector_type(4)));
typedef int tn_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_bf16(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tn_bf16x8, a), __builtin_bit_cast(tn_bf16x8, b), c, 0, 0, 0);
}
// (low half, high half) = (bf16(x0), bf16(x1)), round to nearest even.
// The result must NEVER be an MFMA operand directly: an MFMA that reads a register an asm vector instruction wrote fewer
// than two wait states earlier gets the register's OLD contents, and hipcc puts only `s_nop 0` behind an asm statement
// (tools/valu_mfma_hazard_ubench.hip).  Here every piece goes to LDS (`store`) and comes back through a fragment read;
// tools/audit_asm_loads.py check 4 enforces that on the compiled ISA.
__device__ __forceinline__ unsigned tn_cvt_pk_bf16(float x0, float x1) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
    return r;
}
__device__ __forceinline__ void split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = tn_cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);   // exact
    p2 = tn_cvt_pk_bf16(r0, r1);
    r0 = r0 - __uint_as_float(p2 << 16);
    r1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = tn_cvt_pk_bf16(r0, r1);
}
constexpr int kX6BufBytes = 2 * 8 * 3 * kFragBytes;   // (delta, acts) x 8 tiles x 3 pieces x 1 KiB
constexpr int kX6Lds = 2 * kX6BufBytes;

struct X6Frag {
    f32x4 v;
    template <int OFF>
    __device__ __forceinline__ void issue(uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory"); }
};
// at most N of this wave's LDS operations are still outstanding => the three fragments named are valid
template <int N>
__device__ __forceinline__ void x6_retire(X6Frag (&f)[3]) {
    static_assert(N <= 15, "lgkmcnt is a 4-bit field");
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f[0].v), "+v"(f[1].v), "+v"(f[2].v) : "n"(N) : "memory");
}

// One 256 x 256 product (one split of the points) of the batch below.
#ifdef IDN_DIAG_X6   // diagnostic build only: where a chunk of the bf16-piece dW GEMM spends its cycles (tools/diag_tn_x6.py)
__device__ unsigned long long g_x6_diag[8];   // row 0, rows 1..3 up to the barrier, barrier, tail, prologue + epilogue, chunks, workgroups
// s_memtime returns through lgkmcnt, which the kernel counts by hand for its LDS reads: the stamp waits for itself on the spot,
// at points where no counted read is younger than what the code is about to wait for anyway
#define X6_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#else
#define X6_STAMP(v)
#endif
__device__ __forceinline__ void gemm_tn_x6_item(const TNArgs& g, const int split, char* x6_smem) {
#ifdef IDN_DIAG_X6
    unsigned long long dx_t0 = 0, dx_a = 0, dx_b = 0, dx_c = 0, dx_d = 0, dx_row0 = 0, dx_rows = 0, dx_bar = 0, dx_tail = 0, dx_loop0 = 0, dx_loop1 = 0;
    X6_STAMP(dx_t0);
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const long c_begin = (long)split * g.chunks_per_split;
    long c_end = c_begin + g.chunks_per_split;
    const long c_total = g.P / kTnRows;
    if (c_end > c_total) c_end = c_total;
    const int n_chunks = (int)(c_end - c_begin);          // >= 1 (run_tn_partials sizes the splits so)
    // Raw buffer descriptors based at this split's first row (32-bit offsets span one split).  The loads are inline asm
    // with the destination tied to the register that held the same row of the chunk before ("+v"): as a builtin the
    // reload got a fresh register and a copy at the loop end -- behind a wait for the load, a chunk early.  vmcnt is
    // therefore counted by hand: loads are issued in ONE order (delta rows 0..15, activation rows 0..15) everywhere,
    // so when pair j is split, exactly 30 loads are younger than its second row.
    // (With cache-hot reloads -- -DIDN_X6_TIMING_SAME_ROWS -- the kernel is 8 % faster; a second register set per matrix -- two
    // chunks in flight, vmcnt(62) -- was built and measured in round 4: no gain, so it is not the prefetch depth.  profiles/HISTORY.md)
    auto make_rsrc = [](const float* ptr) {
        const uint64_t a64 = (uint64_t)(uintptr_t)ptr;
        return tn_i32x4{(int)(uint32_t)a64, (int)((uint32_t)(a64 >> 32) & 0xffffu), (int)0xfffffffcu, 0x00020000};
    };
    const tn_i32x4 rsrcA = make_rsrc(g.A + c_begin * kTnRows * (long)g.lda);
    const tn_i32x4 rsrcB = make_rsrc(g.B + c_begin * kTnRows * (long)g.ldb);
    const int voff = tid * 4;
    const int voffB = g.b_split ? (tid < 128 ? g.b_off0 : g.b_off1) + (tid & 127) * 4 : voff;
    const int rowA = g.lda * 4, rowB = g.ldb * 4;
    float ra[kTnRows], rb[kTnRows];
#pragma unroll
    for (int p = 0; p < kTnRows; ++p) ra[p] = rb[p] = 0.f;
    auto load_row_at = [&](float& dst, const tn_i32x4& rsrc, int soff, int vo) {
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(soff) : "memory");
    };
    auto load_row = [&](float& dst, const tn_i32x4& rsrc, int soff) { load_row_at(dst, rsrc, soff, voff); };
    auto load_chunk = [&](int rc) {   // chunk rc of this split, clamped to its last one (re-read, never used)
        const int base = (rc < n_chunks ? rc : n_chunks - 1) * kTnRows;
#pragma unroll
        for (int p = 0; p < kTnRows; ++p) load_row(ra[p], rsrcA, (base + p) * rowA);
#pragma unroll
        for (int p = 0; p < kTnRows; ++p) load_row_at(rb[p], rsrcB, (base + p) * rowB, voffB);
    };
    // this thread's slot: tile tid / 32, lane (tid % 32, hh) -> hh-th half of the fragment
    char* const my_slot = x6_smem + (tid >> 5) * (3 * kFragBytes) + (tid & 31) * 16;
    auto split_store = [&](int buf, int X, const float (&r)[kTnRows]) {
        unsigned pw[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(r[2 * j], r[2 * j + 1], pw[0][j], pw[1][j], pw[2][j]);
        char* dst = my_slot + buf * kX6BufBytes + X * (8 * 3 * kFragBytes);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
                *reinterpret_cast<tn_u32x4*>(dst + q * kFragBytes + h2 * 512) =
                    tn_u32x4{pw[q][4 * h2], pw[q][4 * h2 + 1], pw[q][4 * h2 + 2], pw[q][4 * h2 + 3]};
    };
    const bool do_colsum = g.cpart != nullptr;
    float csum = 0.0f, csum1 = 0.0f;
    auto colsum = [&]() {   // chunk 0 (the loop adds the others as it splits them)
        float s0 = (ra[0] + ra[1]) + (ra[2] + ra[3]), s1 = (ra[4] + ra[5]) + (ra[6] + ra[7]);
        float s2 = (ra[8] + ra[9]) + (ra[10] + ra[11]), s3 = (ra[12] + ra[13]) + (ra[14] + ra[15]);
        csum += (s0 + s1) + (s2 + s3);
    };
    const uint32_t smem0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)x6_smem;
    const uint32_t a_base = smem0 + (4 * wr) * (3 * kFragBytes) + lane * 16;
    const uint32_t b_base = smem0 + (8 + 4 * wc) * (3 * kFragBytes) + lane * 16;

    load_chunk(0);
#pragma unroll
    for (int p = 0; p < kTnRows; ++p) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[p]), "+v"(rb[p])::"memory");
    colsum();
    split_store(0, 0, ra);
    split_store(0, 1, rb);
    load_chunk(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // The fragments of a chunk's FIRST tile row (fa[0], fb[0], fb[1]: nine reads) are issued one chunk EARLY: the chunk barrier
    // sits behind MFMA 59 of the 72 of rows 1..3 -- by then this wave's pieces of the next chunk are stored (slices end at MFMA
    // 51) and fa[0] / fb[0] / fb[1] have had their last use -- so the next chunk's first reads run under the last twelve MFMAs
    // instead of behind a barrier with nothing to overlap (IDN_X6_EARLY_BARRIER=0: the barrier at the end of the chunk).
#ifndef IDN_X6_EARLY_BARRIER
#define IDN_X6_EARLY_BARRIER 1
#endif
    X6Frag fa[4][3], fb[4][3];
    auto issue_first_row = [&](uint32_t pa, uint32_t pb) {
        tn_static_for<3>([&](auto Q) { fa[0][decltype(Q)::value].template issue<decltype(Q)::value * kFragBytes>(pa); });
        tn_static_for<2>([&](auto Y) {
            tn_static_for<3>([&](auto Q) {
                fb[decltype(Y)::value][decltype(Q)::value].template issue<(decltype(Y)::value * 3 + decltype(Q)::value) * kFragBytes>(pb);
            });
        });
    };
    if (IDN_X6_EARLY_BARRIER) issue_first_row(a_base, b_base);
    X6_STAMP(dx_loop0);
#pragma unroll 1
    for (int rc = 0; rc < n_chunks; ++rc) {
        X6_STAMP(dx_a);
        const int buf = rc & 1;
        const uint32_t pa = a_base + buf * kX6BufBytes, pb = b_base + buf * kX6BufBytes;
        // issue order = consumption order: row 0 (nine of its reads are already in flight), then the other delta tiles
        if (!IDN_X6_EARLY_BARRIER) issue_first_row(pa, pb);
        tn_static_for<2>([&](auto Y) {
            tn_static_for<3>([&](auto Q) {
                fb[decltype(Y)::value + 2][decltype(Q)::value].template issue<((decltype(Y)::value + 2) * 3 + decltype(Q)::value) * kFragBytes>(pb);
            });
        });
        tn_static_for<3>([&](auto X) {
            tn_static_for<3>([&](auto Q) {
                fa[decltype(X)::value + 1][decltype(Q)::value].template issue<((decltype(X)::value + 1) * 3 + decltype(Q)::value) * kFragBytes>(pa);
            });
        });
        auto mm2 = [&](int x, int y0, int y1) {   // two tile pairs, interleaved: six piece products each
#define X6_TERM(QA, QB)                                                   \
    acc[x][y0] = mfma_bf16(fa[x][QA].v, fb[y0][QB].v, acc[x][y0]);        \
    acc[x][y1] = mfma_bf16(fa[x][QA].v, fb[y1][QB].v, acc[x][y1]);
            X6_TERM(0, 0) X6_TERM(0, 1) X6_TERM(1, 0) X6_TERM(1, 1) X6_TERM(0, 2) X6_TERM(2, 0)
#undef X6_TERM
        };
        // row 0: 24 reads are outstanding; fa[0], fb[0], fb[1] are the oldest nine
        x6_retire<15>(fa[0]);
        x6_retire<15>(fb[0]);
        x6_retire<15>(fb[1]);
        mm2(0, 0, 1);
        x6_retire<9>(fb[2]);
        x6_retire<9>(fb[3]);
        mm2(0, 2, 3);
        x6_retire<0>(fa[1]);
        x6_retire<0>(fa[2]);
        x6_retire<0>(fa[3]);
        X6_STAMP(dx_b);
        // Rows 1..3: 72 MFMAs, each followed by one SLICE of the side work (the next chunk's split + store, the reloads
        // with the chunk after it) and a scheduling fence: at most ~6 vector instructions, two loads or one store behind
        // an MFMA that occupies the pipe for 32 cycles.  (Left alone, hipcc issues 40 MFMAs back to back and then 50
        // vector and memory instructions in a row, during which the matrix pipe runs dry: 57 % busy.)
        const unsigned live_mask = rc + 1 < n_chunks ? 0xffffffffu : 0u;   // the clamped re-read of the last chunk does not count
#ifdef IDN_X6_TIMING_SAME_ROWS   // timing-only experiment (wrong results): every reload re-reads this split's first chunk (cache hits)
        const int nbase = 0;
#else
        const int nbase = (rc + 2 < n_chunks ? rc + 2 : n_chunks - 1) * kTnRows;   // chunk rc + 2, clamped (re-read, never used)
#endif
        char* const dst = my_slot + (buf ^ 1) * kX6BufBytes;
        unsigned pw[2][3][8];
        float t0 = 0.f, t1 = 0.f;
        // Timing-only experiments (WRONG results), -DIDN_X6_TIMING_DROP=<bits>: 1 drops the split arithmetic of the slices, 2 their LDS
        // stores, 4 their reloads -- which part of the side work costs the matrix pipe its idle cycles (profiles/HISTORY.md, round 4)
#ifndef IDN_X6_TIMING_DROP
#define IDN_X6_TIMING_DROP 0
#endif
        auto slice = [&](auto X_, auto S_) {
            constexpr int X = decltype(X_)::value, sl = decltype(S_)::value;
            float (&r)[kTnRows] = *(X ? &rb : &ra);
            auto store = [&](auto Q_, auto H_) {
                constexpr int q = decltype(Q_)::value, h2 = decltype(H_)::value;
                if constexpr (IDN_X6_TIMING_DROP & 2) return;
                *reinterpret_cast<tn_u32x4*>(dst + X * (8 * 3 * kFragBytes) + q * kFragBytes + h2 * 512) =
                    tn_u32x4{pw[X][q][4 * h2], pw[X][q][4 * h2 + 1], pw[X][q][4 * h2 + 2], pw[X][q][4 * h2 + 3]};
            };
            if constexpr (sl < 24) {
                constexpr int j = sl / 3, ph = sl % 3;
                if constexpr (ph == 0 && (IDN_X6_TIMING_DROP & 1)) {
                    if constexpr (!(IDN_X6_TIMING_DROP & 4)) asm volatile("s_waitcnt vmcnt(30)" : "+v"(r[2 * j]), "+v"(r[2 * j + 1])::"memory");
                    pw[X][0][j] = __float_as_uint(r[2 * j]);
                    pw[X][1][j] = __float_as_uint(r[2 * j + 1]);
                    pw[X][2][j] = __float_as_uint(r[2 * j]);
                } else if constexpr (ph == 1 && (IDN_X6_TIMING_DROP & 1)) {
                } else if constexpr (ph == 0) {
                    // rows 2 j, 2 j + 1 of the chunk being split have landed: 30 younger loads may still be in flight
                    if constexpr (!(IDN_X6_TIMING_DROP & 4)) asm volatile("s_waitcnt vmcnt(30)" : "+v"(r[2 * j]), "+v"(r[2 * j + 1])::"memory");
                    const unsigned p1 = tn_cvt_pk_bf16(r[2 * j], r[2 * j + 1]);
                    pw[X][0][j] = p1;
                    t0 = r[2 * j] - __uint_as_float(p1 << 16);
                    t1 = r[2 * j + 1] - __uint_as_float(p1 & 0xffff0000u);
                    if constexpr (X == 0) {   // (masked, not multiplied: packed-fp32 forms would tie the reload registers to aligned pairs)
                        csum += __uint_as_float(__float_as_uint(r[2 * j]) & live_mask);
                        csum1 += __uint_as_float(__float_as_uint(r[2 * j + 1]) & live_mask);
                        // pinned here: hipcc otherwise keeps the 16 values for one batch of adds at the end, i.e. copies every
                        // row register before its reload (and copies in-flight registers back at the loop end)
                        asm volatile("" : "+v"(csum), "+v"(csum1));
                    }
                } else if constexpr (ph == 1) {
                    const unsigned p2 = tn_cvt_pk_bf16(t0, t1);
                    pw[X][1][j] = p2;
                    t0 = t0 - __uint_as_float(p2 << 16);
                    t1 = t1 - __uint_as_float(p2 & 0xffff0000u);
                    pw[X][2][j] = tn_cvt_pk_bf16(t0, t1);
                } else {
                    if constexpr (!(IDN_X6_TIMING_DROP & 4)) {
                        load_row_at(r[2 * j], X ? rsrcB : rsrcA, (nbase + 2 * j) * (X ? rowB : rowA), X ? voffB : voff);
                        load_row_at(r[2 * j + 1], X ? rsrcB : rsrcA, (nbase + 2 * j + 1) * (X ? rowB : rowA), X ? voffB : voff);
                    }
                    if constexpr (j >= 3 && j < 6) store(ic_<j - 3>{}, ic_<0>{});
                    if constexpr (j == 7) store(ic_<0>{}, ic_<1>{});
                }
            } else if constexpr (sl == 24) {
                store(ic_<1>{}, ic_<1>{});
            } else if constexpr (sl == 25) {
                store(ic_<2>{}, ic_<1>{});
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        tn_static_for<72>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            constexpr int x = 1 + k / 24, kk = k % 24, y = 2 * (kk / 12) + (kk % 2), t = (kk % 12) / 2;
            constexpr int qa = t == 2 || t == 3 ? 1 : (t == 5 ? 2 : 0);   // a1 b1, a1 b2, a2 b1, a2 b2, a1 b3, a3 b1
            constexpr int qb = t == 1 || t == 3 ? 1 : (t == 4 ? 2 : 0);
            acc[x][y] = mfma_bf16(fa[x][qa].v, fb[y][qb].v, acc[x][y]);
            constexpr int kX1 = IDN_X6_EARLY_BARRIER ? 26 : 36;     // where the activation matrix's slices start (each matrix has 26)
            if constexpr (k < kX1) slice(ic_<0>{}, ic_<k>{});
            else slice(ic_<1>{}, ic_<k - kX1>{});
            if constexpr (IDN_X6_EARLY_BARRIER && k == 59) {
                // everyone's pieces of chunk rc + 1 are in LDS; everyone has read chunk rc's (all 24 fragment reads were retired
                // before row 1): the chunk barrier, twelve MFMAs early, and behind it the first row of the next chunk
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                X6_STAMP(dx_c);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                X6_STAMP(dx_d);
#ifdef IDN_DIAG_X6
                dx_row0 += dx_b - dx_a;
                dx_rows += dx_c - dx_b;
                dx_bar += dx_d - dx_c;
#endif
                issue_first_row(a_base + (buf ^ 1) * kX6BufBytes, b_base + (buf ^ 1) * kX6BufBytes);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (!IDN_X6_EARLY_BARRIER) {
            // everyone's pieces of chunk rc + 1 are in LDS; everyone has read chunk rc's
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#ifdef IDN_DIAG_X6
        X6_STAMP(dx_a);            // (the first-row reads of the next chunk, issued twelve MFMAs ago, are waited for here)
        dx_tail += dx_a - dx_d;
        dx_loop1 = dx_a;
#endif
    }
    if (IDN_X6_EARLY_BARRIER) {   // the first-row reads issued behind the last chunk's barrier (of a chunk that does not exist): retired, never used
        x6_retire<0>(fa[0]);
        x6_retire<0>(fb[0]);
        x6_retire<0>(fb[1]);
    }
    // the clamped re-reads of the last chunk are still in flight into ra / rb: retired here, so that the next item of a batch
    // starts on registers nothing is about to write (the stores below then drain under that item's first loads)
#pragma unroll
    for (int p = 0; p < kTnRows; ++p) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[p]), "+v"(rb[p])::"memory");
    if (do_colsum) g.cpart[(long)split * g.N + tid] = csum + csum1;
    // lane (i, hh), tile (x, y), register r: output (32 (4 wr + x) + d_row(r, hh), 32 (4 wc + y) + i)
    float* out = g.part + (long)split * g.N * g.K;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(long)(32 * (4 * wr + x) + d_row(r, hh)) * g.K + 32 * (4 * wc + y) + i] = acc[x][y][r];
#ifdef IDN_DIAG_X6
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long dx_t1;
    X6_STAMP(dx_t1);
    if (tid == 0) {
        atomicAdd(&g_x6_diag[0], dx_row0);
        atomicAdd(&g_x6_diag[1], dx_rows);
        atomicAdd(&g_x6_diag[2], dx_bar);
        atomicAdd(&g_x6_diag[3], dx_tail);
        atomicAdd(&g_x6_diag[4], (dx_t1 - dx_t0) - (dx_loop1 - dx_loop0));
        atomicAdd(&g_x6_diag[5], (unsigned long long)n_chunks);
        atomicAdd(&g_x6_diag[6], 1ull);
        atomicAdd(&g_x6_diag[7], dx_t1 - dx_t0);
    }
#endif
}
// The 256 x 256 weight-gradient products of a pass as ONE launch, ONE PRODUCT PER WORKGROUP: workgroup z works on item
// z / splits over split z % splits of the points, with splits = 2 #CUs / #items (56 for the nine products of a pass on 256 CUs).
// Round 3 walked all items in every workgroup (256 splits each): 9 x 256 partial blocks of 256 KB per pass -- 590 MB written
// and read back by the reduction, whatever the number of points (17 % of the coarse pass's traffic).  Now a workgroup keeps
// its accumulators over 1 / 56 of the points and writes ONE block: 131 MB per pass, and the reduction reads less than a quarter.
constexpr int kMaxTnBatch = 12;
struct TNBatch {
    TNArgs it[kMaxTnBatch];
    int n, splits;
};
__global__ __launch_bounds__(256) void gemm_tn_x6_kernel(TNBatch b) {
    extern __shared__ __attribute__((aligned(16))) char x6_smem[];
    const int item = __builtin_amdgcn_readfirstlane((int)blockIdx.z / b.splits);
    const int split = __builtin_amdgcn_readfirstlane((int)blockIdx.z - item * b.splits);
    gemm_tn_x6_item(b.it[item], split, x6_smem);
}

// out[n*ldo + k] = sum_s part[s][n][k],  n < rows, k < cols  (rows/cols may be smaller than N/K: padding dropped)
// 64 outputs x 4 split lanes per block: lane q adds splits q, q+4, ... in fp64, the four lanes are
// then added in a fixed order (deterministic), so 4x as many loads are in flight per output.
// All partial-slab reductions of a pass in ONE launch (they were 27 launches of a few microseconds each per
// pass, each waiting for the one before it).  Block b works on the item whose block range holds b.
constexpr int kMaxReduceItems = 32;
struct ReduceItem {
    const float* part;   // [splits][N][K], already offset to the first row / column wanted
    float* out;
    int splits, N, K, ldo, rows, cols;
    int block_end;       // exclusive end of this item's block range
};
struct ReduceBatch {
    ReduceItem it[kMaxReduceItems];
    int n;
};
__global__ __launch_bounds__(256) void reduce_batch_kernel(ReduceBatch b) {
    __shared__ double red[4][64];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.it[i].block_end) ++i;   // block-uniform
    const ReduceItem& t = b.it[i];
    const int block0 = i ? b.it[i - 1].block_end : 0;
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = ((int)blockIdx.x - block0) * 64 + o;
    const bool live = idx < t.rows * t.cols;
    const int n = live ? idx / t.cols : 0, k = live ? idx % t.cols : 0;
    double s = 0.0;
    if (live) {
        const float* src = t.part + (long)n * t.K + k;
        const long stride = (long)t.N * t.K;
#pragma unroll 8
        for (int sp = q; sp < t.splits; sp += 4) s += (double)src[sp * stride];
    }
    red[q][o] = s;
    __syncthreads();
    if (q == 0 && live) t.out[(long)n * t.ldo + k] = (float)(((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]);
}

// ---------------------------------------------------------------------------
// compositing backward (baseline.py:325-375 differentiated; cumprod's gradient as
// PyTorch computes it when no factor is zero: reverse cumsum(grad * out) / input).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double shfl_up_dd(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_dd(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

struct CompBwdArgs {
    const float4* raw; const float* z; const float* rays; const float* bc;
    const float* g_rgb;  // [n,3] d rgb_map (may be null)
    const float* g_fg;   // [n,3] d rgb_fg  (may be null)
    const float* g_lw;   // [n]   d last_weight (may be null)
    const float* g_acc;  // [n]   d acc_map (may be null)
    float* d_rgb; int ld_rgb;    // row p = ray*S+s: d raw_rgb at d_rgb[p*ld_rgb + 0..2]
    float* d_sig; int ld_sig;    // d raw_sigma at d_sig[p*ld_sig]
    long n_rays; int S;
};

template <int SPL>
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.n_rays) return;
    const int S = a.S;
    const float* rr = a.rays + ray * IDN_RAY_FLOATS;
    const float dn = sqrtf((rr[3] * rr[3] + rr[4] * rr[4]) + rr[5] * rr[5]);
    const float4* rawr = a.raw + ray * S;
    const float* zr = a.z + ray * S;
    float gr = 0, gg = 0, gb = 0, fr = 0, fg = 0, fb = 0, glw = 0, gacc = 0;
    if (a.g_rgb) { gr = a.g_rgb[ray * 3]; gg = a.g_rgb[ray * 3 + 1]; gb = a.g_rgb[ray * 3 + 2]; }
    if (a.g_fg) { fr = a.g_fg[ray * 3]; fg = a.g_fg[ray * 3 + 1]; fb = a.g_fg[ray * 3 + 2]; }
    if (a.g_lw) glw = a.g_lw[ray];
    if (a.g_acc) gacc = a.g_acc[ray];

    float zs[SPL + 1];
    float4 rw[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        zs[i] = s < S ? zr[s] : 0.f;
        rw[i] = s < S ? rawr[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    zs[SPL] = __shfl_down(zs[0], 1, 64);
    float alpha[SPL], tf[SPL], ex[SPL], dist[SPL];
    double local = 1.0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        float d = (s >= S - 1) ? 1e10f : (zs[i + 1] - zs[i]);
        d = d * dn;
        const float e = expf(-(fmaxf(rw[i].w, 0.0f) + 1e-6f) * d);
        dist[i] = d;
        ex[i] = e;
        alpha[i] = (s < S) ? 1.0f - e : 0.0f;
        tf[i] = (s < S) ? (1.0f - (1.0f - e)) + 1e-10f : 1.0f;
        local *= (double)tf[i];
    }
    // exclusive prefix product over lanes
    double incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_dd(incl, d);
        if (lane >= d) incl *= o;
    }
    double run = shfl_up_dd(incl, 1);
    if (lane == 0) run = 1.0;
    float T[SPL], w[SPL], dw[SPL], cr[SPL], cg[SPL], cb[SPL];
    double lsum = 0.0;  // sum of dw*w over this lane's samples
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        T[i] = (float)run;
        w[i] = alpha[i] * T[i];
        run *= (double)tf[i];
        dw[i] = 0.f;
        cr[i] = cg[i] = cb[i] = 0.f;
        if (s < S) {
            if (s == S - 1) {
                cr[i] = a.bc[ray * 3]; cg[i] = a.bc[ray * 3 + 1]; cb[i] = a.bc[ray * 3 + 2];
                dw[i] = (gr * cr[i] + gg * cg[i] + gb * cb[i]) + glw + gacc;
            } else {
                cr[i] = 1.0f / (1.0f + expf(-rw[i].x));
                cg[i] = 1.0f / (1.0f + expf(-rw[i].y));
                cb[i] = 1.0f / (1.0f + expf(-rw[i].z));
                dw[i] = ((gr + fr) * cr[i] + (gg + fg) * cg[i] + (gb + fb) * cb[i]) + gacc;
            }
            lsum += (double)dw[i] * (double)w[i];
        }
    }
    // suffix sums of dw*w: total - inclusive prefix
    double pin = lsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_dd(pin, d);
        if (lane >= d) pin += o;
    }
    double total = lsum;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) total += shfl_xor_dd(total, m);
    double before = pin - lsum;  // sum over earlier lanes
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        if (s < S) {
            before += (double)dw[i] * (double)w[i];          // inclusive up to s
            const double suffix = total - before;            // sum_{t > s} dw_t w_t
            const float dalpha = dw[i] * T[i] - (float)(suffix / (double)tf[i]);
            const float dsig = (rw[i].w > 0.f) ? dalpha * ex[i] * dist[i] : 0.f;
            const long p = ray * S + s;
            a.d_sig[p * a.ld_sig] = dsig;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            if (s < S - 1) {
                d0 = w[i] * (gr + fr) * cr[i] * (1.0f - cr[i]);
                d1 = w[i] * (gg + fg) * cg[i] * (1.0f - cg[i]);
                d2 = w[i] * (gb + fb) * cb[i] * (1.0f - cb[i]);
            }
            a.d_rgb[p * a.ld_rgb + 0] = d0;
            a.d_rgb[p * a.ld_rgb + 1] = d1;
            a.d_rgb[p * a.ld_rgb + 2] = d2;
        }
    }
}

// ---------------------------------------------------------------------------
// conditioning fold backward
//   dW0[:, 63+c] = db0'[n] cond[c];  dW5[:, 63+c] = db5'[n] cond[c];  dWv0[:, 283+e] = dbv'[n] expr3[e]
//   d cond[c] = sum_n W0[n][63+c] db0'[n] + W5[n][63+c] db5'[n]   -> d aud, d latent (accumulated)
// ---------------------------------------------------------------------------
struct FoldBwdArgs {
    idn_facenerf_params p;
    const float* aud; const float* expr; const float* latent;
    const float* db0; const float* db5; const float* dbv;  // [256], [256], [128]
    float* gW0; float* gW5; float* gWv0;                    // gradient tensors (full nn.Linear layout)
    float* d_aud; float* d_latent;                          // accumulated (+=), may be null
};
__device__ __forceinline__ float cond_val(const FoldBwdArgs& d, int c) {
    if (c < d.p.dim_aud) return d.aud[c];
    c -= d.p.dim_aud;
    if (c < d.p.dim_expr) return d.expr[c] * 1.0f / 3.0f;
    c -= d.p.dim_expr;
    return d.latent[c];
}
__global__ void fold_bwd_kernel(FoldBwdArgs d) {
    const int C = d.p.dim_aud + d.p.dim_expr + d.p.dim_latent;
    const int ld0 = IDN_PTS_CH + C, ld5 = IDN_PTS_CH + C + IDN_W, ldv = IDN_W + IDN_VIEWS_CH + d.p.dim_expr;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_outer = IDN_W * C;
    if (idx < n_outer) {
        const int n = idx / C, c = idx % C;
        const float cv = cond_val(d, c);
        d.gW0[(long)n * ld0 + IDN_PTS_CH + c] = d.db0[n] * cv;
        d.gW5[(long)n * ld5 + IDN_PTS_CH + c] = d.db5[n] * cv;
    } else if (idx < n_outer + (IDN_W / 2) * d.p.dim_expr) {
        const int k = idx - n_outer, n = k / d.p.dim_expr, e = k % d.p.dim_expr;
        d.gWv0[(long)n * ldv + IDN_W + IDN_VIEWS_CH + e] = d.dbv[n] * (d.expr[e] * 1.0f / 3.0f);
    } else {
        // d cond[c]: one wavefront per conditioning column, the lanes split the 256 rows (a thread per column
        // walked them as one chain of dependent strided loads: 65 us)
        const int rel = idx - n_outer - (IDN_W / 2) * d.p.dim_expr;   // n_outer and 128 dim_expr are multiples of 64
        const int c = rel >> 6, lane = rel & 63;
        if (c >= C) return;
        double s = 0.0;
        for (int n = lane; n < IDN_W; n += 64)
            s += (double)d.p.pts_w[0][(long)n * ld0 + IDN_PTS_CH + c] * (double)d.db0[n] +
                 (double)d.p.pts_w[5][(long)n * ld5 + IDN_PTS_CH + c] * (double)d.db5[n];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += shfl_xor_dd(s, m);
        if (lane != 0) return;
        if (c < d.p.dim_aud) {
            if (d.d_aud) d.d_aud[c] += (float)s;
        } else if (c >= d.p.dim_aud + d.p.dim_expr) {
            if (d.d_latent) d.d_latent[c - d.p.dim_aud - d.p.dim_expr] += (float)s;
        }
    }
}

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr int kMaxSplits = 256;
constexpr int kColsumBlocks = 256;   // rows of the column-sum partial buffer (>= kMaxSplits)
constexpr int kGemmsPerPass = 16;
constexpr int kX6ItemsPerPass = 9;   // pts_linears.1..7, views_linears.0 + alpha_linear, the views_linears.1 / .2 pair
// partial-slab floats of one split over all GEMMs of a pass: 9 of 256x256 (pts_linears.1..7, views_linears.0 + alpha_linear, the
// views_linears.1 / .2 pair), 2 of 256x64, 1 of 128x64, 1 of 64x128 -- with room for the fp32 pipe's two 128x128
constexpr size_t kPartFloatsPerSplit = 9 * 65536 + 6 * 16384 + 4 * 8192;   // (the fp32 pipe: 8 + 2 x 16384 for the views pair; the narrow shapes may run 2 x kMaxSplits splits)
// The narrow shapes (256 x 64, 128 x 64, 64 x 128 outputs: pts_linears.0, the encoding columns of pts_linears.5, the direction
// columns of views_linears.0, rgb_linear) are HBM-shaped: ring depth and workgroups per CU decide how many bytes they keep in flight
#ifndef IDN_TN_NARROW_BUFS
#define IDN_TN_NARROW_BUFS 3
#endif
// points per chunk of the narrow shapes: 256 x 64 (40 KiB per 32-point chunk) and 128 x 64 / 64 x 128 (48 KiB per 64-point chunk)
#ifndef IDN_TN_ROWS_4x1
#define IDN_TN_ROWS_4x1 32
#endif
#ifndef IDN_TN_ROWS_THIN
#define IDN_TN_ROWS_THIN 64
#endif
#ifndef IDN_TN_NARROW_SPLITS
#define IDN_TN_NARROW_SPLITS 256
#endif

struct BwdWs {
    float *dA[8], *dV[2], *dV0, *dRGB, *part, *cpart, *wbwd;
    size_t bytes;
};
static BwdWs carve_bwd(char* base, int64_t p_pad) {
    BwdWs w;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = reinterpret_cast<float*>(base + off);
        off += al256(floats * 4);
        return p;
    };
    for (int l = 0; l < 8; ++l) w.dA[l] = take((size_t)p_pad * 256);  // delta of pts_linears.l (pre-activation)
    // delta of views_linears.2 | delta of views_linears.1 side by side in ONE 256-column matrix: their two 128 x 128 weight
    // gradients are then the diagonal blocks of a single 256 x 256 product (launch_pass_bwd)
    w.dV[0] = take((size_t)p_pad * 256);
    w.dV[1] = w.dV[0] + 128;
    w.dV0 = take((size_t)p_pad * 256);
    w.dRGB = take((size_t)p_pad * 64);
    w.part = take((size_t)kMaxSplits * kPartFloatsPerSplit);   // one slab per GEMM of the pass: they are all reduced at its end
    w.cpart = take((size_t)kColsumBlocks * 256 * kGemmsPerPass);
    w.wbwd = take(IDN_DELTA_X6 ? bwd_stream_floats_x6() : (size_t)kBwdStreamFrags * kFragFloats);   // transposed weight stream of the delta chain
    w.bytes = off;
    return w;
}

size_t bwd_workspace_bytes(int64_t n_points) {
    const int64_t p_pad = (n_points + 127) / 128 * 128;
    return carve_bwd(nullptr, p_pad).bytes;
}

// part[split][N][K] = A[:, :N]^T . B[:, :K] over point splits; returns the split count
enum { kPipeX6 = 0, kPipeF32 = 1 };   // which matrix pipe a 256 x 256 GEMM runs on (the other shapes: fp32)
// The backward's pipe for the delta chain and the 256 x 256 GEMMs: the six-piece bf16 arithmetic unless the process was
// started with IDN_BACKWARD_PIPE=f32 (read once; the A/B arm and the fallback, exercised by the GPU tests) or the
// library was built with -DIDN_DELTA_X6=0 / -DIDN_DW_X6=0.
static int env_pipe_f32() {
    static const int v = [] {
        const char* e = getenv("IDN_BACKWARD_PIPE");
        return (e && e[0] == 'f' && e[1] == '3' && e[2] == '2' && e[3] == 0) ? 1 : 0;
    }();
    return v;
}
static int default_gemm_pipe() { return (IDN_DW_X6 && !env_pipe_f32()) ? kPipeX6 : kPipeF32; }
// The 256 x 256 bf16-piece products of a pass, collected and launched as one kernel (gemm_tn_x6_kernel walks them)
struct X6Pending {
    TNBatch b;
    int splits = 0;
    int expected_items = 1;   // how many products the caller will queue: the CUs are divided among them (run_tn_partials)
    X6Pending() { b.n = 0; }
    int launch(hipStream_t s) {
        if (b.n == 0) return IDN_OK;
        static LaunchSetup setup6;
        int num_cu = 0;
        if (int e = setup6.get([]() -> int {
                IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_x6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kX6Lds));
                return IDN_OK;
            }, &num_cu))
            return e;
        {
            ProfScope prof(s, b.it[0].P * b.n, IDN_PROF_DW_GEMM_X6);
            b.splits = splits;
            hipLaunchKernelGGL(gemm_tn_x6_kernel, dim3(1, 1, splits * b.n), dim3(256), kX6Lds, s, b);
        }
        IDN_HIP_CHECK(hipGetLastError());
        b.n = 0;
        return IDN_OK;
    }
    int add(const TNArgs& g, int item_splits, hipStream_t s) {
        if (b.n && (b.n == kMaxTnBatch || item_splits != splits || g.P != b.it[0].P))   // (a pass's products all have the pass's points: one batch)
            if (int e = launch(s)) return e;
        splits = item_splits;
        b.it[b.n++] = g;
        return IDN_OK;
    }
};
static int device_cus(int* out) {
    static LaunchSetup st;
    return st.get([]() -> int { return IDN_OK; }, out);
}
template <int NTW, int KTW, int NB, int R = kTnRows>
static int launch_tn(const TNArgs& g, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)NB * R * (size_t)(64 * NTW + 64 * KTW) * 4;
    static_assert(lds <= 160 * 1024, "the chunk ring must fit a CU's LDS");
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<NTW, KTW, NB, R>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            return IDN_OK;
        }, &num_cu))
        return e;
    hipLaunchKernelGGL((gemm_tn_kernel<NTW, KTW, NB, R>), grid, dim3(256), lds, s, g);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}
static int run_tn_partials(X6Pending& x6, const float* A, int lda, int N, const float* B, int ldb, int K, int64_t P, float* part,
                           int* splits_out, hipStream_t s, float* cpart = nullptr, int pipe = -1, const float* B2 = nullptr) {
    if (pipe < 0) pipe = default_gemm_pipe();
    // B2: the x6 kernel's split-B form (two 128-column matrices, B for output columns 0..127 and B2 for 128..255)
    if (B2 && !(N == 256 && K == 256 && pipe == kPipeX6)) return fail(IDN_EINVAL, "gemm_tn: a split B needs the 256 x 256 bf16-piece kernel");
    int ntw, ktw;
    if (N == 256 && K == 256) { ntw = 4; ktw = 4; }
    else if (N == 256 && K == 64) { ntw = 4; ktw = 1; }
    else if (N == 128 && K == 256) { ntw = 2; ktw = 4; }   // views_linears.0: its 128 units x the 256 trunk channels
    else if (N == 128 && K == 64) { ntw = 2; ktw = 1; }    //                  ... x the direction encoding
    else if (N == 128 && K == 128) { ntw = 2; ktw = 2; }
    else if (N == 64 && K == 128) { ntw = 1; ktw = 2; }
    else return fail(IDN_EUNSUPPORTED, "gemm_tn: no instantiation for %d x %d", N, K);
    const int bx = N / (64 * ntw), by = K / (64 * ktw);
    const bool narrow = ntw * ktw <= 4 && !(ntw == 2 && ktw == 2);     // <4,1>, <2,1>, <1,2>
    const int rows_per_chunk = !narrow ? kTnRows : (ntw == 4 ? IDN_TN_ROWS_4x1 : IDN_TN_ROWS_THIN);
    if (P % rows_per_chunk) return fail(IDN_EINVAL, "gemm_tn: %lld rows are not a multiple of the %d-row chunk", (long long)P, rows_per_chunk);
    const long chunks = P / rows_per_chunk;
    int splits = (narrow ? IDN_TN_NARROW_SPLITS : kMaxSplits) / (bx * by);
    if (ntw == 4 && ktw == 4 && pipe == kPipeX6) {   // one product per workgroup: the CUs are divided among the pass's products
        int cus = 0;
        if (int e = device_cus(&cus)) return e;
        // two rounds of workgroups per CU: the tail is still balanced (2 x 28 x 9 = 504 workgroups on 256 CUs), the fp32 accumulators
        // run over 1 / 56 of the points (10 500 of the fine pass's 589 824) and the partial blocks are 131 MB per pass
        splits = 2 * cus / (x6.expected_items > 0 ? x6.expected_items : 1);
        if (splits > kMaxSplits) splits = kMaxSplits;
    }
    if (splits > chunks) splits = (int)chunks;
    if (splits < 1) splits = 1;
    const int cps = (int)((chunks + splits - 1) / splits);
    splits = (int)((chunks + cps - 1) / cps);
    TNArgs g{A, lda, B, ldb, part, N, K, (long)P, cps, cpart, 0, 0, 0};
    if (B2) {   // byte offsets from the lower of the two addresses (buffer offsets are unsigned)
        const float* base = B < B2 ? B : B2;
        const int64_t o0 = (int64_t)(B - base) * 4, o1 = (int64_t)(B2 - base) * 4;
        if (o0 >= (int64_t)1 << 31 || o1 >= (int64_t)1 << 31) return fail(IDN_EUNSUPPORTED, "gemm_tn: split B matrices more than 2 GiB apart");
        g.B = base;
        g.b_split = 1;
        g.b_off0 = (int)o0;
        g.b_off1 = (int)o1;
    }
    const dim3 grid(bx, by, splits);
    if (ntw == 4 && ktw == 4 && pipe == kPipeX6) {   // queued: launched with the pass's other 256 x 256 products (ReduceQueue::flush)
        if (int e = x6.add(g, splits, s)) return e;
        *splits_out = splits;
        return IDN_OK;
    }
    ProfScope prof(s, P, IDN_PROF_DW_GEMM);
    int e;
    if (ntw == 4 && ktw == 4) e = launch_tn<4, 4, kTnBufs>(g, grid, s);
    else if (ntw == 4 && ktw == 1) e = launch_tn<4, 1, IDN_TN_NARROW_BUFS, IDN_TN_ROWS_4x1>(g, grid, s);
    else if (ntw == 2 && ktw == 4) e = launch_tn<2, 4, kTnBufs>(g, grid, s);
    else if (ntw == 2 && ktw == 1) e = launch_tn<2, 1, IDN_TN_NARROW_BUFS, IDN_TN_ROWS_THIN>(g, grid, s);
    else if (ntw == 2 && ktw == 2) e = launch_tn<2, 2, kTnBufs>(g, grid, s);
    else e = launch_tn<1, 2, IDN_TN_NARROW_BUFS, IDN_TN_ROWS_THIN>(g, grid, s);
    if (e) return e;
    *splits_out = splits;
    return IDN_OK;
}
// The reductions of a pass are queued and launched together (flush) once every GEMM has been issued.
struct ReduceQueue {
    ReduceBatch b;
    int blocks = 0;
    float* part_next;    // slab pools of the workspace
    float* cpart_next;
    X6Pending x6;        // the 256 x 256 bf16-piece products queued so far: one launch, in front of the reductions
    ReduceQueue(float* part_pool, float* cpart_pool, int expected_x6_items = 1) : part_next(part_pool), cpart_next(cpart_pool) {
        b.n = 0;
        x6.expected_items = expected_x6_items;
    }
    // out[(0..rows) x (0..cols)] (ld ldo) = sum over splits of the N x K partial blocks, from row row0 / column col0 on
    int add(const float* part, int splits, int N, int K, int row0, int col0, float* out, int ldo, int rows, int cols) {
        if (b.n >= kMaxReduceItems) return fail(IDN_EUNSUPPORTED, "reduce queue full");
        blocks += (rows * cols + 63) / 64;
        b.it[b.n++] = ReduceItem{part + (size_t)row0 * K + col0, out, splits, N, K, ldo, rows, cols, blocks};
        return IDN_OK;
    }
    int flush(hipStream_t s) {
        if (int e = x6.launch(s)) return e;
        if (b.n == 0) return IDN_OK;
        hipLaunchKernelGGL(reduce_batch_kernel, dim3(blocks), dim3(256), 0, s, b);
        IDN_HIP_CHECK(hipGetLastError());
        b.n = 0;
        blocks = 0;
        return IDN_OK;
    }
};
// GEMM into a fresh slab of the pool; *part_out / *cpart_out are where its partial blocks went
static int run_tn_q(ReduceQueue& q, const float* A, int lda, int N, const float* B, int ldb, int K, int64_t P, int* splits,
                    const float** part_out, const float** cpart_out, bool colsum, hipStream_t s, int pipe, const float* B2 = nullptr) {
    float* part = q.part_next;
    float* cpart = colsum ? q.cpart_next : nullptr;
    if (int e = run_tn_partials(q.x6, A, lda, N, B, ldb, K, P, part, splits, s, cpart, pipe, B2)) return e;
    q.part_next += (size_t)(*splits) * N * K;
    if (colsum) q.cpart_next += (size_t)(*splits) * N;
    *part_out = part;
    if (cpart_out) *cpart_out = cpart;
    return IDN_OK;
}
// out = A^T B (rows x cols of it); db (optional, `db_cols` entries) = column sums of A, from the same pass over A
static int run_tn(ReduceQueue& q, const float* A, int lda, int N, const float* B, int ldb, int K, int64_t P, float* out,
                  int ldo, int rows, int cols, hipStream_t s, float* db = nullptr, int db_cols = 0,
                  int pipe = -1) {
    int splits = 0;
    const float *part, *cpart;
    if (int e = run_tn_q(q, A, lda, N, B, ldb, K, P, &splits, &part, &cpart, db != nullptr, s, pipe)) return e;
    if (db)
        if (int e = q.add(cpart, splits, 1, N, 0, 0, db, N, 1, db_cols)) return e;
    return q.add(part, splits, N, K, 0, 0, out, ldo, rows, cols);
}

// One 256 x 256 dW GEMM in isolation (idealnerf_dw_gemm: the arithmetic of the bf16-piece GEMM against the fp32 pipe
// and fp64, on inputs a training step does not produce).
size_t dw_gemm_workspace_bytes() { return al256((size_t)kMaxSplits * 65536 * 4) + al256((size_t)kColsumBlocks * 256 * 4); }
int launch_dw_gemm(const float* delta, int ld_delta, const float* acts, int ld_acts, int64_t rows, float* dW, float* db, int pipe,
                   void* ws, size_t ws_bytes, hipStream_t s) {
    if (rows <= 0 || rows % 128) return fail(IDN_EINVAL, "dw_gemm: rows %lld is not a positive multiple of 128", (long long)rows);
    if (ld_delta < 256 || ld_acts < 256) return fail(IDN_EINVAL, "dw_gemm: row pitch < 256");
    const int as_in_a_pass = pipe == 2;   // IDN_DW_PIPE_BF16X6_PASS: the split count a pass runs its nine products with
    if (as_in_a_pass) pipe = kPipeX6;
    if (pipe != kPipeX6 && pipe != kPipeF32) return fail(IDN_EINVAL, "dw_gemm: pipe %d", pipe);
    if (!ws || ws_bytes < dw_gemm_workspace_bytes()) return fail(IDN_EWORKSPACE, "dw_gemm workspace %zu < %zu", ws_bytes, dw_gemm_workspace_bytes());
    float* part = reinterpret_cast<float*>(ws);
    float* cpart = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + al256((size_t)kMaxSplits * 65536 * 4));
    ReduceQueue q(part, cpart, as_in_a_pass ? kX6ItemsPerPass : 1);
    if (int e = run_tn(q, delta, ld_delta, 256, acts, ld_acts, 256, rows, dW, 256, 256, 256, s, db, db ? 256 : 0, pipe)) return e;
    return q.flush(s);
}

int launch_pass_bwd(const idn_facenerf_params& p, const idn_facenerf_grads& gr, const float* aud, const float* expr,
                    const float* latent, const float* acts, const float* raw, const float* z, const float* rays,
                    const float* bc, int64_t n_rays, int S, const float* g_rgb, const float* g_fg, const float* g_lw,
                    const float* g_acc, float* d_aud, float* d_latent, void* ws_, size_t ws_bytes, hipStream_t s) {
    const int64_t P = n_rays * S;
    const int64_t Pp = (P + 127) / 128 * 128;
    if (S < 2 || S > 256) return fail(IDN_EUNSUPPORTED, "pass_bwd: n_samples %d outside [2, 256]", S);
    const BwdWs w = carve_bwd(reinterpret_cast<char*>(ws_), Pp);
    if (!ws_ || ws_bytes < w.bytes) return fail(IDN_EWORKSPACE, "backward workspace %zu < %zu", ws_bytes, w.bytes);
    const int C = p.dim_aud + p.dim_expr + p.dim_latent;
    const int ld0 = IDN_PTS_CH + C, ld5 = IDN_PTS_CH + C + IDN_W, ldv = IDN_W + IDN_VIEWS_CH + p.dim_expr;
    auto act = [&](int i) { return acts + (size_t)act_off(i) * Pp; };
    auto a_l = [&](int l) { return act(kActA1 + l - 1); };  // post-ReLU output of pts_linears.(l-1), l = 1..8
    auto v_l = [&](int l) { return act(kActV1 + l - 1); };  // post-ReLU output of views_linears.(l-1), l = 1..3

    // d(outputs) -> d raw, written straight into the head deltas (zero elsewhere)
    IDN_HIP_CHECK(hipMemsetAsync(w.dRGB, 0, (size_t)Pp * 64 * 4, s));
    // dV0: columns 0..127 are written by the delta chain for every row, column 128 (d sigma) by the compositing
    // backward for every real point; only the padding rows of that column need zeros.  Columns 129..255 are never written
    // here, but the views_linears.0 + alpha_linear product below READS all 256 columns: what it finds there is whatever an
    // earlier pass left in this workspace (the host layer hands over a workspace that was zeroed when it was allocated), and
    // it reaches only output rows / column sums 129..255, which `q.add` never takes
    if (Pp > P) IDN_HIP_CHECK(hipMemsetAsync(w.dV0 + (size_t)P * 256, 0, (size_t)(Pp - P) * 256 * 4, s));
    {
        CompBwdArgs a{reinterpret_cast<const float4*>(raw), z, rays, bc, g_rgb, g_fg, g_lw, g_acc,
                      w.dRGB, 64, w.dV0 + kSigmaChannel, 256, (long)n_rays, S};
        const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
        switch ((S + 63) / 64) {
            case 1: hipLaunchKernelGGL(composite_bwd_kernel<1>, grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL(composite_bwd_kernel<2>, grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL(composite_bwd_kernel<3>, grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL(composite_bwd_kernel<4>, grid, block, 0, s, a); break;
        }
        IDN_HIP_CHECK(hipGetLastError());
    }
#define TRY(x) do { if (int e_ = (x)) return e_; } while (0)
    // All pre-activation deltas in one fused pass over the points (mlp_f32_bwd.hip): dV[0] = delta of
    // views_linears.2, dV[1] = views_linears.1, dV0[:, :128] = views_linears.0 (col 128 = d sigma), dA[l] = pts_linears.l
    if (IDN_DELTA_X6 && !env_pipe_f32()) {
        TRY(launch_pack_bf16x6_bwd(p, w.wbwd, s));
        TRY(launch_delta_chain_x6(w.wbwd, acts, Pp, w.dRGB, w.dV0, w.dV[0], w.dV[1], w.dA, s));
    } else {   // (the workspace's stream buffer is sized for the larger of the two streams)
        TRY(launch_pack_f32_bwd(p, w.wbwd, s));
        TRY(launch_delta_chain(w.wbwd, acts, Pp, w.dRGB, w.dV0, w.dV[0], w.dV[1], w.dA, s));
    }
    // weight and bias gradients: dW_l = delta_l^T a_{l-1} (contraction over the points), db_l = column sums
    ReduceQueue q(w.part, w.cpart, kX6ItemsPerPass);
    TRY(run_tn(q, w.dRGB, 64, 64, v_l(3), 128, 128, Pp, gr.rgb_w, 128, 3, 128, s, gr.rgb_b, 3));
    // views_linears.2 and .1 (128 x 128 each): on the bf16 pipe ONE 256 x 256 launch whose A is the side-by-side delta matrix and
    // whose B columns come from the two activation matrices (v2 | v1) -- the weight gradients are its diagonal blocks, and 1 KB per
    // point and layer is read once by a kernel that runs at the HBM rate (two fp32-MFMA launches took 1.7x as long).  The
    // off-diagonal blocks are computed and dropped.
    const int pipe = default_gemm_pipe();
    if (pipe == kPipeX6) {
        int splits = 0;
        const float *part, *cpart;
        TRY(run_tn_q(q, w.dV[0], 256, 256, v_l(2), 128, 256, Pp, &splits, &part, &cpart, true, s, pipe, v_l(1)));
        TRY(q.add(cpart, splits, 1, 256, 0, 0, gr.views_b[2], 128, 1, 128));
        TRY(q.add(cpart, splits, 1, 256, 0, 128, gr.views_b[1], 128, 1, 128));
        TRY(q.add(part, splits, 256, 256, 0, 0, gr.views_w[2], 128, 128, 128));
        TRY(q.add(part, splits, 256, 256, 128, 128, gr.views_w[1], 128, 128, 128));
    } else {
        TRY(run_tn(q, w.dV[0], 256, 128, v_l(2), 128, 128, Pp, gr.views_w[2], 128, 128, 128, s, gr.views_b[2], 128));
        TRY(run_tn(q, w.dV[1], 256, 128, v_l(1), 128, 128, Pp, gr.views_w[1], 128, 128, 128, s, gr.views_b[1], 128));
    }
    // views_linears.0 and alpha_linear against a8 in ONE 256 x 256 product: A = the 256-column matrix dV0, whose columns 0..127
    // are views_linears.0's deltas and whose column 128 is d sigma (alpha_linear's delta); columns 129..255 are never written --
    // whatever they hold only reaches output rows 129..255, which are not read.  (It replaces a 128 x 256 fp32-MFMA product and
    // a weighted column sum that read dV0 and a8 once each.)  The direction-encoding columns stay a product of their own.
    {
        int splits = 0;
        const float *part, *cpart;
        TRY(run_tn_q(q, w.dV0, 256, 256, a_l(8), 256, 256, Pp, &splits, &part, &cpart, true, s, pipe));
        TRY(q.add(cpart, splits, 1, 256, 0, 0, gr.views_b[0], 128, 1, 128));
        TRY(q.add(cpart, splits, 1, 256, 0, kSigmaChannel, gr.alpha_b, 1, 1, 1));
        TRY(q.add(part, splits, 256, 256, 0, 0, gr.views_w[0], ldv, 128, 256));
        TRY(q.add(part, splits, 256, 256, kSigmaChannel, 0, gr.alpha_w, 256, 1, 256));
    }
    TRY(run_tn(q, w.dV0, 256, 128, act(kActDir), 64, 64, Pp, gr.views_w[0] + IDN_W, ldv, 128, IDN_VIEWS_CH, s));
    for (int l = 7; l >= 1; --l) {
        const float* cur = w.dA[l];
        if (l == 5) {
            TRY(run_tn(q, cur, 256, 256, a_l(5), 256, 256, Pp, gr.pts_w[5] + IDN_PTS_CH + C, ld5, 256, 256, s, gr.pts_b[5], 256));
            TRY(run_tn(q, cur, 256, 256, act(kActX0), 64, 64, Pp, gr.pts_w[5], ld5, 256, IDN_PTS_CH, s));
        } else {
            TRY(run_tn(q, cur, 256, 256, a_l(l), 256, 256, Pp, gr.pts_w[l], 256, 256, 256, s, gr.pts_b[l], 256));
        }
    }
    TRY(run_tn(q, w.dA[0], 256, 256, act(kActX0), 64, 64, Pp, gr.pts_w[0], ld0, 256, IDN_PTS_CH, s, gr.pts_b[0], 256));
    TRY(q.flush(s));
#undef TRY
    {
        FoldBwdArgs f{p, aud, expr, latent, gr.pts_b[0], gr.pts_b[5], gr.views_b[0], gr.pts_w[0], gr.pts_w[5],
                      gr.views_w[0], d_aud, d_latent};
        const int total = IDN_W * C + (IDN_W / 2) * p.dim_expr + C * 64;   // one wavefront per conditioning column at the end
        if (total > 0) {
            hipLaunchKernelGGL(fold_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, f);
            IDN_HIP_CHECK(hipGetLastError());
        }
    }
    return IDN_OK;
}

#ifdef IDN_DIAG_X6
extern "C" int idealnerf_diag_x6_read(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x6_diag), 8 * sizeof(unsigned long long)) != hipSuccess) return -3;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_x6_diag), z, sizeof(z)) != hipSuccess) return -3;
    return 0;
}
#endif
#ifdef IDN_DIAG
extern "C" int idealnerf_diag_tn_read(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tn_diag), 8 * sizeof(unsigned long long)) != hipSuccess) return -3;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tn_diag), z, sizeof(z)) != hipSuccess) return -3;
    return 0;
}
#endif

}  // namespace idn
