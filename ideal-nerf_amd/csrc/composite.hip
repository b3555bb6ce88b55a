// Per-ray stages around the MLP (gfx950): ray generation, coarse depths, alpha
// compositing, inverse-CDF importance sampling + merge.  One 64-lane wavefront per ray;
// prefix products / sums are wave-level scans in fp64 (PyTorch-CPU's cumprod / cumsum
// accumulate in double and round each output to fp32 -- DESIGN.md "numerics").
//
// Built with -ffp-contract=off: every product and sum below rounds separately, as the
// reference's chain of eager ops does; the sample positions feed index decisions.
#include "idn_internal.h"

namespace idn {

constexpr int kMaxSpl = 4;  // samples per lane: S <= 256

__device__ __forceinline__ double shfl_up_d(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_d(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    return v;
}
// inclusive scans across the 64 lanes
__device__ __forceinline__ double wave_scan_mul_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_d(v, d);
        if (lane >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ double wave_scan_add_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_d(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// ---------------------------------------------------------------------------
// a1: get_rays + record assembly (helper.py:228-243, audio_exp_nerf.py:396-427)
// ---------------------------------------------------------------------------
struct C2W {
    float m[12];
};
__global__ void frame_rays_kernel(C2W c, int W, float focal, float cx, float cy, float near_, float far_, int row0,
                                  int npix, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const int row = row0 + idx / W, col = idx % W;
    const float i = (float)col, j = (float)row;
    const float d0 = (i - cx) / focal;
    const float d1 = -(j - cy) / focal;
    const float d2 = -1.0f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = (d0 * c.m[4 * r + 0] + d1 * c.m[4 * r + 1]) + d2 * c.m[4 * r + 2];
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    float* o = out + (long)idx * IDN_RAY_FLOATS;
    o[0] = c.m[3];
    o[1] = c.m[7];
    o[2] = c.m[11];
    o[3] = d[0];
    o[4] = d[1];
    o[5] = d[2];
    o[6] = near_;
    o[7] = far_;
    o[8] = d[0] / nrm;
    o[9] = d[1] / nrm;
    o[10] = d[2] / nrm;
}

int launch_frame_rays(const float* c2w_host, int H, int W, float focal, float cx, float cy, float near_, float far_,
                      int row0, int nrows, float* rays_out, hipStream_t s) {
    C2W c;
    for (int i = 0; i < 12; ++i) c.m[i] = c2w_host[i];
    if (cx < 0) cx = W * 0.5f;
    if (cy < 0) cy = H * 0.5f;
    const int npix = nrows * W;
    hipLaunchKernelGGL(frame_rays_kernel, dim3((npix + 255) / 256), dim3(256), 0, s, c, W, focal, cx, cy, near_, far_,
                       row0, npix, rays_out);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

// ---------------------------------------------------------------------------
// a3: coarse depths (audio_exp_nerf.py:306-330)
// ---------------------------------------------------------------------------
__global__ void coarse_depths_kernel(const float* rays, const float* t_vals, const float* t_rand, long n_rays, int S,
                                     int lindisp, float* z) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rays * S) return;
    const long r = idx / S;
    const int s = (int)(idx - r * S);
    const float near_ = rays[r * IDN_RAY_FLOATS + 6], far_ = rays[r * IDN_RAY_FLOATS + 7];
    auto zlin = [&](int k) {
        const float t = t_vals[k];
        if (lindisp) return 1.0f / (1.0f / near_ * (1.0f - t) + 1.0f / far_ * t);   // linear in inverse depth (:309-310)
        return near_ * (1.0f - t) + far_ * t;
    };
    float zz = zlin(s);
    if (t_rand) {
        const float lower = (s == 0) ? zz : 0.5f * (zz + zlin(s - 1));
        const float upper = (s == S - 1) ? zz : 0.5f * (zlin(s + 1) + zz);
        const float tr = (s == S - 1) ? 1.0f : t_rand[idx];
        zz = lower + (upper - lower) * tr;
    }
    z[idx] = zz;
}

int launch_coarse_depths(const float* rays, const float* t_vals, const float* t_rand, int64_t n_rays, int S,
                         int lindisp, float* z, hipStream_t s) {
    const long total = (long)n_rays * S;
    if (total <= 0) return IDN_OK;
    hipLaunchKernelGGL(coarse_depths_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rays, t_vals,
                       t_rand, (long)n_rays, S, lindisp, z);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

// ---------------------------------------------------------------------------
// frame tail: to8b (NeRFs/HeadNeRF/helper.py:154, `(255 * np.clip(x, 0, 1)).astype(np.uint8)`) and
// the NaN/Inf scan of the render dict (audio_exp_nerf.py:367-369) as ONE device-side flag.
// 255 * clip(x) is an fp32 product, astype truncates.  A NaN pixel is written as 0 and flagged.
// ---------------------------------------------------------------------------
__global__ void to8b_kernel(const float* __restrict__ rgb, long n_values, int swap_rb, unsigned char* __restrict__ out,
                            int* __restrict__ flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n_values) {
        const float x = rgb[i];
        bad = !(fabsf(x) <= 3.402823466e+38f);  // NaN or +-Inf
        float c = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
        if (x != x) c = 0.0f;
        const long px = i / 3;
        const int ch = (int)(i - px * 3);
        out[px * 3 + (swap_rb ? 2 - ch : ch)] = (unsigned char)(int)(255.0f * c);
    }
    if (flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

int launch_to8b(const float* rgb, int64_t n_pixels, int swap_rb, unsigned char* out, int* flag, hipStream_t s) {
    const long n = (long)n_pixels * 3;
    if (n <= 0) return IDN_OK;
    hipLaunchKernelGGL(to8b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rgb, n, swap_rb, out, flag);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

// ---------------------------------------------------------------------------
// a6: raw2outputs (NeRFs/HeadNeRF/train/baseline.py:325-375; rgb_fg: TorsoNeRF/run_nerf.py:757)
// Lane l owns samples l*SPL .. l*SPL+SPL-1 (contiguous, so a ray's prefix product is a
// lane-local product followed by one wave scan).
// ---------------------------------------------------------------------------
// One ray per wave.  wout[i] = weight of sample lane * SPL + i (0 beyond S), for a caller that goes on with them.
template <int SPL>
__device__ __forceinline__ void composite_ray(const float4* raw, const float* z, const float* rays, const float* bc,
                                              long ray, int lane, int S, const float* noise, int white_bkgd,
                                              const idn_composite_out& out, float (&wout)[SPL]) {
    const float* rr = rays + ray * IDN_RAY_FLOATS;
    const float dn = sqrtf((rr[3] * rr[3] + rr[4] * rr[4]) + rr[5] * rr[5]);  // torch.norm(rays_d)
    const float4* rawr = raw + ray * S;
    const float* zr = z + ray * S;

    float zs[SPL + 1];
    float4 rw[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        const bool ok = s < S;
        zs[i] = ok ? zr[s] : 0.f;
        rw[i] = ok ? rawr[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (noise && ok) rw[i].w = rw[i].w + noise[ray * S + s];   // raw_noise_std: drawn by the caller (baseline.py:353-361)
    }
    zs[SPL] = __shfl_down(zs[0], 1, 64);  // first sample of the next lane

    float alpha[SPL], tf[SPL];
    double local = 1.0;  // product of this lane's (1 - alpha + 1e-10)
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        float dist = (s >= S - 1) ? 1e10f : (zs[i + 1] - zs[i]);
        dist = dist * dn;
        const float a = 1.0f - expf(-(fmaxf(rw[i].w, 0.0f) + 1e-6f) * dist);
        alpha[i] = (s < S) ? a : 0.0f;
        tf[i] = (s < S) ? (1.0f - a) + 1e-10f : 1.0f;
        local *= (double)tf[i];
    }
    const double incl = wave_scan_mul_d(local, lane);
    double run = shfl_up_d(incl, 1);  // exclusive prefix over lanes
    if (lane == 0) run = 1.0;

    double sr = 0, sg = 0, sb = 0, sd = 0, sw = 0, fr = 0, fg = 0, fb = 0;
    float wlast = 0.f;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        const float T = (float)run;  // cumprod output, rounded to fp32 per element
        const float w = alpha[i] * T;
        run *= (double)tf[i];
        wout[i] = (s < S) ? w : 0.0f;
        if (s < S) {
            float cr, cg, cb;
            if (s == S - 1) {  // last sample's colour := background pixel (baseline.py:352)
                cr = bc[ray * 3 + 0];
                cg = bc[ray * 3 + 1];
                cb = bc[ray * 3 + 2];
                wlast = w;
            } else {
                cr = 1.0f / (1.0f + expf(-rw[i].x));
                cg = 1.0f / (1.0f + expf(-rw[i].y));
                cb = 1.0f / (1.0f + expf(-rw[i].z));
                fr += (double)(w * cr);
                fg += (double)(w * cg);
                fb += (double)(w * cb);
            }
            sr += (double)(w * cr);
            sg += (double)(w * cg);
            sb += (double)(w * cb);
            sd += (double)(w * zs[i]);
            sw += (double)w;
            if (out.weights) out.weights[ray * S + s] = w;
        }
    }
    sr = wave_sum_d(sr); sg = wave_sum_d(sg); sb = wave_sum_d(sb);
    sd = wave_sum_d(sd); sw = wave_sum_d(sw);
    if (out.rgb_fg) { fr = wave_sum_d(fr); fg = wave_sum_d(fg); fb = wave_sum_d(fb); }
    if (out.last_weight) {
        // the lane owning sample S-1 holds it
        const int owner = (S - 1) / SPL;
        const float lw = __shfl(wlast, owner, 64);
        if (lane == 0) out.last_weight[ray] = lw;
    }
    if (lane == 0) {
        const float depth = (float)sd, acc = (float)sw;
        if (out.rgb_map) {
            const float white = white_bkgd ? 1.0f - acc : 0.0f;   // rgb_map + (1 - acc_map) (baseline.py:372-373)
            out.rgb_map[ray * 3 + 0] = white_bkgd ? (float)sr + white : (float)sr;
            out.rgb_map[ray * 3 + 1] = white_bkgd ? (float)sg + white : (float)sg;
            out.rgb_map[ray * 3 + 2] = white_bkgd ? (float)sb + white : (float)sb;
        }
        if (out.rgb_fg) {
            out.rgb_fg[ray * 3 + 0] = (float)fr;
            out.rgb_fg[ray * 3 + 1] = (float)fg;
            out.rgb_fg[ray * 3 + 2] = (float)fb;
        }
        if (out.depth_map) out.depth_map[ray] = depth;
        if (out.acc_map) out.acc_map[ray] = acc;
        if (out.disp_map) out.disp_map[ray] = 1.0f / fmaxf(1e-10f, depth / acc);
    }
}

template <int SPL>
__global__ __launch_bounds__(256) void composite_kernel(const float4* raw, const float* z, const float* rays,
                                                        const float* bc, long n_rays, int S, const float* noise,
                                                        int white_bkgd, idn_composite_out out) {
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;  // wave-uniform
    float w[SPL];
    composite_ray<SPL>(raw, z, rays, bc, ray, lane, S, noise, white_bkgd, out, w);
}

int launch_composite(const float* raw, const float* z, const float* rays, const float* bc, int64_t n_rays, int S,
                     const float* noise, int white_bkgd, const idn_composite_out& out, hipStream_t s) {
    if (n_rays <= 0) return IDN_OK;
    if (S < 2 || S > 64 * kMaxSpl) return fail(IDN_EUNSUPPORTED, "composite: n_samples %d outside [2, %d]", S, 64 * kMaxSpl);
    const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
    const float4* r4 = reinterpret_cast<const float4*>(raw);
    const int spl = (S + 63) / 64;
    switch (spl) {
        case 1: hipLaunchKernelGGL(composite_kernel<1>, grid, block, 0, s, r4, z, rays, bc, (long)n_rays, S, noise, white_bkgd, out); break;
        case 2: hipLaunchKernelGGL(composite_kernel<2>, grid, block, 0, s, r4, z, rays, bc, (long)n_rays, S, noise, white_bkgd, out); break;
        case 3: hipLaunchKernelGGL(composite_kernel<3>, grid, block, 0, s, r4, z, rays, bc, (long)n_rays, S, noise, white_bkgd, out); break;
        default: hipLaunchKernelGGL(composite_kernel<4>, grid, block, 0, s, r4, z, rays, bc, (long)n_rays, S, noise, white_bkgd, out); break;
    }
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

// ---------------------------------------------------------------------------
// a7 + a8: sample_pdf (helper.py:269-313) and sorted merge (audio_exp_nerf.py:347-349)
//
// Per ray (one wave): bins / cdf staged in LDS, inverse CDF by binary search
// (searchsorted right=True: inds = #{k : cdf[k] <= u}), merge of the two depth lists by rank
// (values only are kept, so any total order gives torch.sort's values).
// ---------------------------------------------------------------------------
constexpr int kMaxBins = 256;   // S - 1 <= 255
constexpr int kMaxNi = 256;
constexpr int kMaxFine = 512;

struct SampleArgs {
    const float* z;        // [n,S] coarse depths (null when bins_in is given)
    const float* weights;  // [n,S] (the kernel uses [:,1:-1]) or, with bins_in, [n,nb-1] as helper.sample_pdf takes them
    const float* cdf_in;   // [n,nb] optional: skip the pdf/cdf stage (bit-exact boundary)
    const float* bins_in;  // [n,nb]
    const float* u;
    int u_per_ray;
    long n_rays;
    int S, Ni, nb;
    float* z_samples;
    int64_t* inds;
    float* cdf_out;
    float* z_fine;
    float* z_std;
};

// torch.sum(x, -1) of a contiguous fp32 row as PyTorch's CPU kernel evaluates it (ATen
// native/cpu/SumKernel.cpp: vectorized_inner_sum -> row_sum -> multi_row_sum, the AVX2 build that
// is dispatched on AVX2 and AVX512 hosts alike): 8-lane vectors; vector i goes to accumulator i&3
// while i < 4*(nv/4), the remaining vectors to accumulator 0; accumulators 1..3 are added to 0 in
// turn; the scalar tail x[8*nv..] is summed from zero, then the 8 lanes are added one by one.
// multi_row_sum only starts cascading at 16 rows of 4 vectors (K >= 512), above the sizes taken here.
// Rows shorter than one vector (K < 8) take ATen's scalar row_sum instead: element i goes to partial
// sum i&3 while i < 4*(K/4), the rest to partial sum 0, then partial sums 1..3 are added to 0 in turn.
// (Both forms checked against torch.sum on the build host for K = 1..513: 100 % bit-identical.)
// This is the sum that normalises the pdf (helper.py:272) and therefore decides importance indices:
// reproducing its order makes cdf and inds bit-identical to the reference for identical weights
// (tests/golden/sample_pdf.npz, frame32.npz).  w: this wave's row in LDS.
__device__ __forceinline__ float aten_row_sum(const float* w, int K, int lane) {
    if (K < 8) {
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        const int n4 = K >> 2;
        for (int i = 0; i < n4; ++i) {
            p0 = p0 + w[4 * i];
            p1 = p1 + w[4 * i + 1];
            p2 = p2 + w[4 * i + 2];
            p3 = p3 + w[4 * i + 3];
        }
        for (int k = n4 * 4; k < K; ++k) p0 = p0 + w[k];
        return ((p0 + p1) + p2) + p3;
    }
    const int nv = K >> 3, ni = nv >> 2;
    const int acc_id = (lane >> 3) & 3, j = lane & 7;
    float acc = 0.0f;
    for (int i = 0; i < ni; ++i) acc = acc + w[((i * 4 + acc_id) << 3) + j];
    if (acc_id == 0)
        for (int i = ni * 4; i < nv; ++i) acc = acc + w[(i << 3) + j];
    const float a1 = __shfl(acc, j + 8, 64), a2 = __shfl(acc, j + 16, 64), a3 = __shfl(acc, j + 24, 64);
    acc = ((acc + a1) + a2) + a3;  // meaningful in lanes 0..7
    float total = 0.0f;
    for (int k = nv << 3; k < K; ++k) total = total + w[k];
#pragma unroll
    for (int q = 0; q < 8; ++q) total = total + __shfl(acc, q, 64);
    return total;
}

// Ascending total order with NaN last (torch.sort's): a before b?
__device__ __forceinline__ bool sort_lt(float a, float b) { return a < b || (b != b && a == a); }
__device__ __forceinline__ bool sort_eq(float a, float b) { return a == b || (a != a && b != b); }

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS writes of this wave are done
}

// One ray per wave; cdf / bins / val are this wave's LDS rows.  `w_lds` (optional): the ray's S compositing
// weights already in LDS (the fused march kernel), instead of a.weights in global memory.
__device__ __forceinline__ void sample_pdf_ray(const SampleArgs& a, long ray, int lane, float* cdf, float* bins, float* val,
                                               const float* w_lds) {
    const int nb = a.nb;

    if (a.cdf_in) {
        for (int k = lane; k < nb; k += 64) {
            cdf[k] = a.cdf_in[ray * nb + k];
            bins[k] = a.bins_in[ray * nb + k];
        }
    } else {
        const int np = nb - 1;
        const float* wr;
        if (a.bins_in) {  // helper.sample_pdf(bins, weights, ...): the caller's own bins and weights[n, nb-1]
            for (int k = lane; k < nb; k += 64) bins[k] = a.bins_in[ray * nb + k];
            wr = a.weights + ray * np;
        } else {          // bins = z midpoints; weights[:, 1:-1] (audio_exp_nerf.py:340-342)
            const float* zr = a.z + ray * a.S;
            for (int k = lane; k < nb; k += 64) bins[k] = 0.5f * (zr[k + 1] + zr[k]);
            wr = (w_lds ? w_lds : a.weights + ray * a.S) + 1;
        }
        // w' = w + 1e-5; lane l owns pdf entries l*4 .. l*4+3
        float wp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane * 4 + i;
            wp[i] = (k < np) ? wr[k] + 1e-5f : 0.0f;
            if (k < np) val[k] = wp[i];
        }
        wave_lds_fence();
        const float total = aten_row_sum(val, np, lane);  // torch.sum(weights, -1, keepdim=True)
        double pl = 0.0;
        float pdf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pdf[i] = wp[i] / total;
            pl += (double)pdf[i];
        }
        const double incl = wave_scan_add_d(pl, lane);
        double run = incl - pl;  // exclusive prefix (sum of earlier lanes)
        if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane * 4 + i;
            run += (double)pdf[i];
            if (k < np) cdf[k + 1] = (float)run;  // cumsum output rounded per element
        }
    }
    wave_lds_fence();
    if (a.cdf_out)
        for (int k = lane; k < nb; k += 64) a.cdf_out[ray * nb + k] = cdf[k];

    // ---- inverse CDF
    double m1 = 0.0;
    float zsv[kMaxNi / 64];
#pragma unroll
    for (int ii = 0; ii < kMaxNi / 64; ++ii) {
        const int i = ii * 64 + lane;
        zsv[ii] = 0.f;
        if (i < a.Ni) {
            const float u = a.u_per_ray ? a.u[ray * a.Ni + i] : a.u[i];
            int lo = 0, hi = nb;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= u) lo = mid + 1;
                else hi = mid;
            }
            const int below = max(0, lo - 1), above = min(nb - 1, lo);
            const float cb = cdf[below], ca = cdf[above];
            const float bb = bins[below], ba = bins[above];
            float denom = ca - cb;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (u - cb) / denom;
            const float zsamp = bb + t * (ba - bb);
            zsv[ii] = zsamp;
            m1 += (double)zsamp;
            if (a.inds) a.inds[ray * a.Ni + i] = (int64_t)lo;
            if (a.z_samples) a.z_samples[ray * a.Ni + i] = zsamp;
        }
    }
    if (a.z_std) {  // torch.std(z_samples, unbiased=False)  (audio_exp_nerf.py:363)
        const double mean = wave_sum_d(m1) / (double)a.Ni;
        double m2 = 0.0;
#pragma unroll
        for (int ii = 0; ii < kMaxNi / 64; ++ii)
            if (ii * 64 + lane < a.Ni) {
                const double dlt = (double)zsv[ii] - mean;
                m2 += dlt * dlt;
            }
        m2 = wave_sum_d(m2);
        if (lane == 0) a.z_std[ray] = (float)sqrt(m2 / (double)a.Ni);
    }
    // ---- z_fine = sort(cat[z_coarse, z_samples]).  Element e of the concatenation goes to slot
    // rank(e) = #{j : val[j] before val[e], ties by position}.  Both halves are normally sorted
    // already (coarse depths always; the samples whenever u is sorted, i.e. perturb == 0), and then
    // the rank is the element's own position plus one binary search in the other half; otherwise
    // (random u) every element is counted against all others.  Either way the slots are those of a
    // stable sort, so the output does not depend on which branch ran.
    if (a.z_fine) {
        const int nf = a.S + a.Ni;
        const float* zr = a.z + ray * a.S;
        wave_lds_fence();  // the row sum's reads of val are done
        for (int k = lane; k < a.S; k += 64) val[k] = zr[k];
#pragma unroll
        for (int ii = 0; ii < kMaxNi / 64; ++ii)
            if (ii * 64 + lane < a.Ni) val[a.S + ii * 64 + lane] = zsv[ii];
        wave_lds_fence();
        bool ordered = true;
        for (int e = lane; e < nf; e += 64)
            if (e + 1 < nf && e + 1 != a.S && sort_lt(val[e + 1], val[e])) ordered = false;
        if (__all(ordered)) {
            for (int e = lane; e < nf; e += 64) {
                const float v = val[e];
                const bool first = e < a.S;
                // other half: first-half elements count strictly smaller ones, second-half elements also equal ones
                int lo = first ? a.S : 0, hi = first ? nf : a.S;
                const int base = lo;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const float o = val[mid];
                    const bool before = first ? sort_lt(o, v) : (sort_lt(o, v) || sort_eq(o, v));
                    if (before) lo = mid + 1;
                    else hi = mid;
                }
                const int rank = (first ? e : e - a.S) + (lo - base);
                a.z_fine[ray * nf + rank] = v;
            }
        } else {
            for (int e = lane; e < nf; e += 64) {
                const float v = val[e];
                int rank = 0;
                for (int j = 0; j < nf; ++j) {
                    const float o = val[j];
                    rank += (sort_lt(o, v) || (sort_eq(o, v) && j < e)) ? 1 : 0;
                }
                a.z_fine[ray * nf + rank] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(SampleArgs a) {
    __shared__ float s_cdf[4][kMaxBins];
    __shared__ float s_bins[4][kMaxBins];
    __shared__ float s_val[4][kMaxFine];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * 4 + wv;
    if (ray >= a.n_rays) return;  // wave-uniform; no block-level barrier below
    sample_pdf_ray(a, ray, lane, s_cdf[wv], s_bins[wv], s_val[wv], nullptr);
}

// ---------------------------------------------------------------------------
// The ray march between the two network passes as ONE kernel (audio_exp_nerf.py:335-349): coarse
// raw2outputs, sample_pdf and the sorted merge, one wave per ray.  The compositing weights go from the
// lanes' registers into the wave's LDS row and the pdf / cdf / inversion / merge run on them there: the
// [n, S] weight matrix never exists in HBM (unless its debug tap is asked for), and the coarse pass of a
// render is MLP -> march -> MLP -> composite.
// ---------------------------------------------------------------------------
template <int SPL>
__global__ __launch_bounds__(256) void march_kernel(const float4* raw, const float* rays, const float* bc, const float* noise,
                                                    int white_bkgd, idn_composite_out out, SampleArgs a) {
    __shared__ float s_cdf[4][kMaxBins];
    __shared__ float s_bins[4][kMaxBins];
    __shared__ float s_val[4][kMaxFine];
    __shared__ float s_w[4][kMaxBins + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * 4 + wv;
    if (ray >= a.n_rays) return;  // wave-uniform; no block-level barrier below
    float w[SPL];
    composite_ray<SPL>(raw, a.z, rays, bc, ray, lane, a.S, noise, white_bkgd, out, w);
#pragma unroll
    for (int i = 0; i < SPL; ++i)
        if (lane * SPL + i < a.S) s_w[wv][lane * SPL + i] = w[i];
    wave_lds_fence();
    sample_pdf_ray(a, ray, lane, s_cdf[wv], s_bins[wv], s_val[wv], s_w[wv]);
}

int launch_march(const float* raw, const float* z, const float* rays, const float* bc, const float* noise, int white_bkgd,
                 const idn_composite_out& out, const float* u, int u_per_ray, int64_t n_rays, int S, int Ni,
                 float* z_samples, int64_t* inds, float* cdf_out, float* z_fine, float* z_std, hipStream_t s) {
    if (n_rays <= 0) return IDN_OK;
    const int nb = S - 1;
    if (S < 3 || S > 64 * kMaxSpl || nb > kMaxBins - 1) return fail(IDN_EUNSUPPORTED, "march: n_samples %d outside [3, %d]", S, kMaxBins);
    if (Ni < 1 || Ni > kMaxNi) return fail(IDN_EUNSUPPORTED, "march: n_importance %d outside [1, %d]", Ni, kMaxNi);
    if (S + Ni > kMaxFine) return fail(IDN_EUNSUPPORTED, "march: n_samples + n_importance > %d", kMaxFine);
    SampleArgs a{z, nullptr, nullptr, nullptr, u, u_per_ray, (long)n_rays, S, Ni, nb, z_samples, inds, cdf_out, z_fine, z_std};
    const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
    const float4* r4 = reinterpret_cast<const float4*>(raw);
    switch ((S + 63) / 64) {
        case 1: hipLaunchKernelGGL(march_kernel<1>, grid, block, 0, s, r4, rays, bc, noise, white_bkgd, out, a); break;
        case 2: hipLaunchKernelGGL(march_kernel<2>, grid, block, 0, s, r4, rays, bc, noise, white_bkgd, out, a); break;
        case 3: hipLaunchKernelGGL(march_kernel<3>, grid, block, 0, s, r4, rays, bc, noise, white_bkgd, out, a); break;
        default: hipLaunchKernelGGL(march_kernel<4>, grid, block, 0, s, r4, rays, bc, noise, white_bkgd, out, a); break;
    }
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

int launch_sample_pdf(const float* z, const float* weights, const float* cdf_in, const float* bins_in,
                      const float* u, int u_per_ray, int64_t n_rays, int S, int Ni, float* z_samples,
                      int64_t* inds, float* cdf_out, float* z_fine, float* z_std, hipStream_t s) {
    if (n_rays <= 0) return IDN_OK;
    const int nb = S - 1;
    if (nb < 2 || nb > kMaxBins - 1) return fail(IDN_EUNSUPPORTED, "sample_pdf: %d bins outside [2, %d]", nb, kMaxBins - 1);
    if (Ni < 1 || Ni > kMaxNi) return fail(IDN_EUNSUPPORTED, "sample_pdf: n_importance %d outside [1, %d]", Ni, kMaxNi);
    if (S + Ni > kMaxFine) return fail(IDN_EUNSUPPORTED, "sample_pdf: n_samples + n_importance > %d", kMaxFine);
    if (z_fine && !z) return fail(IDN_EINVAL, "sample_pdf: the merged depths need the coarse depths z");
    SampleArgs a{z, weights, cdf_in, bins_in, u, u_per_ray, (long)n_rays, S, Ni, nb, z_samples, inds, cdf_out, z_fine, z_std};
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
