// The whole per-ray path of Network.render_rays (NeRFs/HeadNeRF/train/audio_exp_nerf.py:300-371) as ONE kernel, fp32:
// coarse network on 64 samples -> raw2outputs -> sample_pdf -> merge -> fine network on 192 samples -> raw2outputs,
// with the sample positions, the raw network outputs, the compositing weights and the cdf of a ray staged in LDS.
// Nothing per-sample crosses HBM: a ray costs its 44-byte record, 256 bytes of coarse depths, 12 bytes of background
// pixel in, and its output pixels out (north_star's "fused ray-march kernel").
//
// A persistent workgroup (one per CU, 4 waves) takes the rays in GROUPS of four:
//   passes 0, 1   the coarse FaceNeRF on the group's 4 x 64 = 256 points (two 128-point tiles of the fp32 MLP kernel's
//                 pass -- fused PE + 12 layers of v_mfma_f32_32x32x2_f32, weights streamed through the LDS ring; raw -> LDS)
//   march         wave w composites ray w, inverts its cdf and merges the depths (march.h: the same device functions as
//                 march_kernel, on LDS rows) -> 192 fine depths of ray w in LDS
//   passes 2..7   the fine FaceNeRF on the group's 4 x 192 = 768 points (depths from LDS; raw -> LDS)
//   composite     wave w composites ray w's 192 fine samples -> the output pixel
// The weight ring alternates between the two packed networks (WStreamT<..., DUAL>: the descriptor is switched where the
// prefetch wraps, so the first slice of the next pass's network is in flight during the last slice of this pass), and the
// folded bias block in LDS is exchanged at the two network changes of a group, its fetch hidden under the march / the compositing.
//
// Same arithmetic as the kernel sequence (mlp_f32_kernel -> march_kernel -> mlp_f32_kernel -> composite_kernel), bit for bit:
// tests/test_hip_parity.py::test_fused_ray_kernel_equals_the_unfused_path.  The same kernel also runs as TWO launches
// (PHASE: coarse network + march | fine network + compositing), one network each, with the fine depths through HBM.
// Measured: one launch is the fastest arrangement on a full frame (1.000-1.005 of the sequence) but moves
// six times its HBM bytes -- two 2.25 MiB streams alternate through a 4 MiB L2 --; two launches move half the sequence's bytes
// at 0.995 of its speed.  The sequence stays the default; these are options (idn_render_args::fused_march = 1 / 2).
#include "march.h"
#include "mlp_f32_layers.h"

namespace idn {

constexpr int kFG = 4;                              // rays per group (one per wave in the march)
constexpr int kFS = 64, kFNi = 128, kFSf = kFS + kFNi;
constexpr int kFCoarsePasses = kFG * kFS / 128;     // 2
constexpr int kFFinePasses = kFG * kFSf / 128;      // 6
constexpr int kFPasses = kFCoarsePasses + kFFinePasses;
constexpr int kFScratchFloats = 64 + 64 + kFSf + 72;   // per wave: cdf, bins, val (pdf terms, then the merge), weights
// LDS: ring | bias block (padded) | fine depths [4][192] | union { raw fine [768] x 16 B ; raw coarse [256] x 16 B + march scratch }
constexpr int kFUnionBytes = kFG * kFSf * 16;
static_assert(kFG * kFS * 16 + 4 * kFScratchFloats * 4 <= kFUnionBytes, "march scratch aliases the fine raw rows");
constexpr int kFBiasPerThread = (kBiasFloats + 255) / 256;       // 13: the block in LDS is padded to a whole number of floats per thread
constexpr int kFBiasPadFloats = kFBiasPerThread * 256;
constexpr int kFusedLds = kRingFrags * kFragBytes + kFBiasPadFloats * 4 + kFG * kFSf * 4 + kFUnionBytes;
static_assert(kFusedLds <= 160 * 1024, "LDS per CU");

struct FusedArgs {
    const float* wstream_c;
    const float* bias_c;
    const float* wstream_f;
    const float* bias_f;
    const float* rays;   // [n, 11]
    const float* bc;     // [n, 3]
    const float* z_c;    // [n, 64] coarse depths (coarse_depths_kernel)
    const float* u;
    int u_per_ray;
    long n_rays;
    int white_bkgd;
    idn_composite_out co, fo;   // outputs of the coarse / fine compositing, indexed by ray
    float* z_std;
    float* tap_raw_c;    // debug taps (any may be null)
    float* tap_raw_f;
    float* tap_z_fine;
    int64_t* tap_inds;
    float* tap_z_samples;
    float* tap_cdf;
    float* z_f;          // [n, 192] fine depths in HBM: written by the coarse half, read by the fine half (split arrangement only)
    Draws draws;         // on: the importance draws come from the Philox table instead of `u`
};

using WStreamDual = WStreamT<4, kSliceFrags, kRingSlots, true>;

// PHASE: kFusedWhole = the whole ray in one launch; kFusedCoarse = coarse network + march (fine depths -> a.z_f);
// kFusedFine = fine network + compositing (fine depths <- a.z_f).  The two halves together are the SPLIT arrangement: one
// network per launch, so the weight stream stays in L2 like in the kernel sequence, and still no raw output, weight or cdf in
// HBM -- only the 768 bytes of fine depths per ray cross between the launches.
enum { kFusedWhole = 0, kFusedCoarse = 1, kFusedFine = 2 };
template <int PHASE>
constexpr int fused_lds() {
    return kRingFrags * kFragBytes + kFBiasPadFloats * 4 + (PHASE == kFusedWhole ? kFG * kFSf * 4 : 0) +
           (PHASE == kFusedCoarse ? kFG * kFS * 16 + 4 * kFScratchFloats * 4 : kFUnionBytes);
}
static_assert(fused_lds<kFusedWhole>() == kFusedLds, "LDS layout");

template <int PHASE>
__global__ __launch_bounds__(256, 1) void render_fused_kernel(FusedArgs a) {
    constexpr bool kWhole = PHASE == kFusedWhole;
    constexpr int kP0 = PHASE == kFusedFine ? kFCoarsePasses : 0;                 // this launch's passes of a group: [kP0, kP1)
    constexpr int kP1 = PHASE == kFusedCoarse ? kFCoarsePasses : kFPasses;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_s = reinterpret_cast<float*>(smem + kRingFrags * kFragBytes);
    float* zf_s = bias_s + kFBiasPadFloats;                                        // (whole-ray arrangement only)
    char* uni = reinterpret_cast<char*>(zf_s + (kWhole ? kFG * kFSf : 0));
    float4* rawf_s = reinterpret_cast<float4*>(uni);
    float4* rawc_s = reinterpret_cast<float4*>(uni);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    float* scratch = reinterpret_cast<float*>(uni + kFG * kFS * 16) + wave * kFScratchFloats;

    Diag dg;
    std::conditional_t<kWhole, WStreamDual, WStream> ws;
    ws.dg = &dg;
    if constexpr (kWhole) ws.init_dual(a.wstream_c, a.wstream_f, kFCoarsePasses, kFPasses, kNumSlices, ring, tid, wave);
    else ws.init(PHASE == kFusedCoarse ? a.wstream_c : a.wstream_f, kNumSlices, ring, tid, wave);
    PeLane pln;
    pln.init(h);
    FragReader fr;
    fr.addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    fr.addr1 = fr.addr0 + 64 * kFragBytes;
    const float* bias_h = bias_s + 4 * h;
    const long ngroups = (a.n_rays + kFG - 1) / kFG;

    // this lane's point in pass p of a group: index q among the group's coarse (p < 2) or fine points
    auto point_q = [&](int p) { return (p < kFCoarsePasses ? p : p - kFCoarsePasses) * 128 + wave * 32 + m; };
    // ray record (and, in a coarse pass, the depth) of this lane's point in pass p of group g, clamped to the last ray
    auto load_in = [&](long g, int p, PointIn& in) {
        const int q = point_q(p);
        const bool fine = p >= kFCoarsePasses;
        const int rl = fine ? q / kFSf : q >> 6;
        long ray = g * kFG + rl;
        if (ray >= a.n_rays) ray = a.n_rays - 1;
        const float* rr = a.rays + ray * IDN_RAY_FLOATS;
        in.f[0] = rr[0]; in.f[1] = rr[1]; in.f[2] = rr[2];
        in.f[3] = rr[3]; in.f[4] = rr[4]; in.f[5] = rr[5];
        in.f[6] = rr[8]; in.f[7] = rr[9]; in.f[8] = rr[10];
        if constexpr (kWhole) in.f[9] = fine ? 0.0f : a.z_c[ray * kFS + (q & 63)];   // (fine depths come from LDS when the pass starts)
        else in.f[9] = fine ? a.z_f[ray * kFSf + q % kFSf] : a.z_c[ray * kFS + (q & 63)];
    };

    // The folded bias block of the network in use sits in LDS (there is no room for both).  At a change of network the other
    // one's 12.5 KB are fetched into registers BEFORE the march / the final compositing and written to LDS AFTER it, behind
    // the barrier those need anyway: the fetch rides under work that does not touch the block (as a plain reload in front of
    // the pass it cost two more barriers and an exposed round trip per change: 0.35 % of the frame).
    // (buffer loads through a descriptor bounded to the block: one lane offset and compile-time steps instead of thirteen
    //  64-bit addresses per lane, and the lanes past the end of the block read zeros into the LDS padding)
    float bias_next[kFBiasPerThread];
    const uint32_t bias_voff = tid * 4;
    auto bias_fetch = [&](const float* src) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, kBiasFloats * 4, 0x00020000);
        static_for<kFBiasPerThread>([&](auto I) {
            bias_next[decltype(I)::value] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, bias_voff, 1024 * decltype(I)::value, 0));
        });
    };
    auto bias_commit = [&]() {
        static_for<kFBiasPerThread>([&](auto I) { bias_s[tid + 256 * decltype(I)::value] = bias_next[decltype(I)::value]; });
    };
    bias_fetch(PHASE == kFusedFine ? a.bias_f : a.bias_c);
    bias_commit();
    __syncthreads();

    PointIn cur, nxt;
    load_in(blockIdx.x, kP0, cur);
    nxt = cur;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
#pragma unroll 1
        for (int pass = kP0; pass < kP1; ++pass) {
            // (the pass number as an opaque scalar: left visible, hipcc peels and specialises the loop on it -- four copies of
            //  the 1 680-MFMA pass body and 21 spilled registers)
            int p = pass;
            asm volatile("" : "+s"(p));
            const bool fine = p >= kFCoarsePasses;
            const int q = point_q(p);
            if constexpr (kWhole)
                if (fine) cur.f[9] = zf_s[q];   // row q / 192, sample q % 192: the rows are contiguous

            // ---- inputs: this lane's half of the 64 point features and 32 direction features
            float pe[8][4], pd[4][4];
            {
                float pt[3], v[3];
                point_of<kModeRays>(cur, pt, v);
                PeAxes axp, axd;
                axp.init(pt, h);
                axd.init(v, h);
                static_for<8>([&](auto G) {
                    static_for<4>([&](auto J) {
                        constexpr int gg = decltype(G)::value, j = decltype(J)::value;
                        pe[gg][j] = pe_slot<8 * gg + j, 10>(axp, pln);
                        if constexpr (gg < 4) pd[gg][j] = pe_slot<8 * gg + j, 4>(axd, pln);
                    });
                });
            }

            // ---- the pass: mlp_f32_kernel's inference pass (mlp_f32_layers.h); the next pass's ray record is loaded after
            //      pts_linears.5's first slice opens and touched one layer later
            float rgb[3], sigma;
            f32_inference_pass(pe, pd, bias_s, bias_h, ws, fr,
                               [&]() {
                                   const bool wraps = p + 1 == kP1;
                                   long gn = wraps ? g + gridDim.x : g;
                                   if (gn >= ngroups) gn = g;   // no next group: a valid, unused address
                                   load_in(gn, wraps ? kP0 : p + 1, nxt);
                               },
                               [&]() { touch_point(nxt); }, rgb, sigma);

            // ---- raw stays on chip (the optional taps copy it out)
            if (h == 0) {
                const float4 o = make_float4(rgb[0], rgb[1], rgb[2], sigma);
                (fine ? rawf_s : rawc_s)[q] = o;
                float* tap = fine ? a.tap_raw_f : a.tap_raw_c;
                const int per_ray = fine ? kFSf : kFS;
                const long ray = g * kFG + q / per_ray;
                if (tap && ray < a.n_rays) *reinterpret_cast<float4*>(tap + (ray * per_ray + q % per_ray) * 4) = o;
            }
            cur = nxt;

            // ---- between the networks: wave w marches ray w of the group
            if (PHASE != kFusedFine && p == kFCoarsePasses - 1) {
                __syncthreads();   // (also: every wave is done with the coarse bias block)
                if constexpr (kWhole) bias_fetch(a.bias_f);
                const long ray = g * kFG + wave;
                float* zrow = kWhole ? zf_s + wave * kFSf : nullptr;   // split arrangement: the fine depths go to a.z_f only
                if (ray < a.n_rays) {   // wave-uniform
                    float* cdf = scratch;
                    float* bins = scratch + 64;
                    float* val = scratch + 128;
                    float* wrow = scratch + 128 + kFSf;
                    float wts[1];
                    composite_ray<1>(rawc_s + wave * kFS, a.z_c + ray * kFS, a.rays, a.bc, ray, lane, kFS, nullptr, a.white_bkgd, a.co, wts);
                    wrow[lane] = wts[0];
                    wave_lds_fence();
                    const SampleArgs sa{a.z_c, nullptr, nullptr, nullptr, a.u, a.u_per_ray, a.n_rays, kFS, kFNi, kFS - 1,
                                        a.tap_z_samples, a.tap_inds, a.tap_cdf, kWhole ? a.tap_z_fine : a.z_f, a.z_std, a.draws};
                    sample_pdf_ray(sa, ray, lane, cdf, bins, val, wrow, zrow);
                } else if constexpr (kWhole) {
                    for (int k = lane; k < kFSf; k += 64) zrow[k] = 0.0f;   // a padding ray: its points are computed and dropped
                }
                if constexpr (kWhole) bias_commit();
                __syncthreads();
            }
            // ---- after the fine passes: wave w composites ray w
            if (PHASE != kFusedCoarse && p == kFPasses - 1) {
                __syncthreads();   // (also: every wave is done with the fine bias block)
                if constexpr (kWhole) bias_fetch(a.bias_c);
                const long ray = g * kFG + wave;
                if (ray < a.n_rays) {
                    float wts[3];
                    composite_ray<3>(rawf_s + wave * kFSf, kWhole ? zf_s + wave * kFSf : a.z_f + ray * kFSf, a.rays, a.bc, ray, lane, kFSf, nullptr,
                                     a.white_bkgd, a.fo, wts);
                }
                if constexpr (kWhole) bias_commit();
                __syncthreads();
            }
        }
    }
    // drain the slice prefetched for a pass that will not happen
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}

int launch_render_fused(int arrangement, const float* packed_c, const float* folded_c, const float* packed_f, const float* folded_f,
                        const float* rays, const float* bc, const float* z_c, float* z_f, const float* u, int u_per_ray, int64_t n_rays,
                        int white_bkgd, const idn_composite_out& co, const idn_composite_out& fo, float* z_std, float* tap_raw_c,
                        float* tap_raw_f, float* tap_z_fine, int64_t* tap_inds, float* tap_z_samples, float* tap_cdf, hipStream_t s,
                        Draws draws) {
    if (n_rays <= 0) return IDN_OK;
    if (arrangement != 1 && arrangement != 2) return fail(IDN_EINVAL, "fused march: arrangement %d (1 = one kernel, 2 = coarse + march | fine + compositing)", arrangement);
    if (arrangement == 2 && !z_f) return fail(IDN_EINVAL, "fused march: the two-launch arrangement needs the fine-depth buffer z_f[n, 192]");
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&render_fused_kernel<kFusedWhole>), hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds<kFusedWhole>()));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&render_fused_kernel<kFusedCoarse>), hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds<kFusedCoarse>()));
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&render_fused_kernel<kFusedFine>), hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds<kFusedFine>()));
            return IDN_OK;
        }, &num_cu))
        return e;
    const int64_t ngroups = (n_rays + kFG - 1) / kFG;
    const int grid = (int)(ngroups < num_cu ? ngroups : num_cu);
    FusedArgs a{packed_c, folded_c, packed_f, folded_f, rays, bc, z_c, u, u_per_ray, (long)n_rays, white_bkgd, co, fo, z_std,
                tap_raw_c, tap_raw_f, tap_z_fine, tap_inds, tap_z_samples, tap_cdf, z_f, draws};
    if (arrangement == 1) {
        ProfScope prof(s, n_rays * (kFS + kFSf), IDN_PROF_MLP_FWD);
        hipLaunchKernelGGL(render_fused_kernel<kFusedWhole>, dim3(grid), dim3(256), fused_lds<kFusedWhole>(), s, a);
    } else {
        {
            ProfScope prof(s, n_rays * kFS, IDN_PROF_MLP_FWD);
            hipLaunchKernelGGL(render_fused_kernel<kFusedCoarse>, dim3(grid), dim3(256), fused_lds<kFusedCoarse>(), s, a);
        }
        ProfScope prof(s, n_rays * kFSf, IDN_PROF_MLP_FWD);
        hipLaunchKernelGGL(render_fused_kernel<kFusedFine>, dim3(grid), dim3(256), fused_lds<kFusedFine>(), s, a);
    }
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
