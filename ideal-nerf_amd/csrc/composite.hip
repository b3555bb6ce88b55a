// Per-ray stages around the MLP (gfx950): ray generation, coarse depths, alpha
// compositing, inverse-CDF importance sampling + merge.  One 64-lane wavefront per ray;
// prefix products / sums are wave-level scans in fp64 (PyTorch-CPU's cumprod / cumsum
// accumulate in double and round each output to fp32 -- DESIGN.md "numerics").
//
// Built with -ffp-contract=off: every product and sum below rounds separately, as the
// reference's chain of eager ops does; the sample positions feed index decisions.
#include "march.h"

namespace idn {

// ---------------------------------------------------------------------------
// a1: get_rays + record assembly (helper.py:228-243, audio_exp_nerf.py:396-427)
// ---------------------------------------------------------------------------
struct C2W {
    float m[12];
};
__global__ void frame_rays_kernel(C2W c, int W, float focal, float cx, float cy, float near_, float far_, long pix0,
                                  int npix, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const long pix = pix0 + idx;      // row-major pixel index in the frame
    const int row = (int)(pix / W), col = (int)(pix % W);
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
                       (long)row0 * W, npix, rays_out);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}
// the records of the pixels [pix0, pix0 + npix) of the frame (row-major), for the frame mode of the render call
int launch_frame_rays_pixels(const float* c2w_host, int H, int W, float focal, float cx, float cy, float near_, float far_,
                             int64_t pix0, int npix, float* rays_out, hipStream_t s) {
    C2W c;
    for (int i = 0; i < 12; ++i) c.m[i] = c2w_host[i];
    if (cx < 0) cx = W * 0.5f;
    if (cy < 0) cy = H * 0.5f;
    if (npix <= 0) return IDN_OK;
    hipLaunchKernelGGL(frame_rays_kernel, dim3((npix + 255) / 256), dim3(256), 0, s, c, W, focal, cx, cy, near_, far_,
                       (long)pix0, npix, rays_out);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

// ---------------------------------------------------------------------------
// a3: coarse depths (audio_exp_nerf.py:306-330)
// ---------------------------------------------------------------------------
__global__ void coarse_depths_kernel(const float* rays, const float* t_vals, const float* t_rand, long n_rays, int S,
                                     int lindisp, float* z, Draws draws) {
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
    if (t_rand || draws.on) {
        const float lower = (s == 0) ? zz : 0.5f * (zz + zlin(s - 1));
        const float upper = (s == S - 1) ? zz : 0.5f * (zlin(s + 1) + zz);
        const float tr = (s == S - 1) ? 1.0f : draws.on ? draw_t_rand(draws, r, s) : t_rand[idx];
        zz = lower + (upper - lower) * tr;
    }
    z[idx] = zz;
}

int launch_coarse_depths(const float* rays, const float* t_vals, const float* t_rand, int64_t n_rays, int S,
                         int lindisp, float* z, hipStream_t s, Draws draws) {
    const long total = (long)n_rays * S;
    if (total <= 0) return IDN_OK;
    hipLaunchKernelGGL(coarse_depths_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rays, t_vals,
                       t_rand, (long)n_rays, S, lindisp, z, draws);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

// The table the in-kernel draws come from, as a tensor (include/idealnerf.h: idealnerf_philox_uniform)
__global__ void philox_uniform_kernel(unsigned long long seed, int which, long row0, long n_rows, int n_cols, float* out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * n_cols) return;
    const long r = idx / n_cols;
    const int c = (int)(idx - r * n_cols);
    out[idx] = philox_uniform(seed, 2ull * (unsigned long long)(row0 + r) + (unsigned long long)which, (unsigned)c);
}
int launch_philox_uniform(unsigned long long seed, int which, int64_t row0, int64_t n_rows, int n_cols, float* out, hipStream_t s) {
    const long total = (long)n_rows * n_cols;
    if (total <= 0) return IDN_OK;
    hipLaunchKernelGGL(philox_uniform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, seed, which, (long)row0, (long)n_rows, n_cols, out);
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

template <int SPL>
__global__ __launch_bounds__(256) void composite_kernel(const float4* raw, const float* z, const float* rays,
                                                        const float* bc, long n_rays, int S, const float* noise,
                                                        int white_bkgd, idn_composite_out out) {
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;  // wave-uniform
    float w[SPL];
    composite_ray<SPL>(raw + ray * S, z + ray * S, rays, bc, ray, lane, S, noise, white_bkgd, out, w);
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
    composite_ray<SPL>(raw + ray * a.S, a.z + ray * a.S, rays, bc, ray, lane, a.S, noise, white_bkgd, out, w);
#pragma unroll
    for (int i = 0; i < SPL; ++i)
        if (lane * SPL + i < a.S) s_w[wv][lane * SPL + i] = w[i];
    wave_lds_fence();
    sample_pdf_ray(a, ray, lane, s_cdf[wv], s_bins[wv], s_val[wv], s_w[wv]);
}

int launch_march(const float* raw, const float* z, const float* rays, const float* bc, const float* noise, int white_bkgd,
                 const idn_composite_out& out, const float* u, int u_per_ray, int64_t n_rays, int S, int Ni,
                 float* z_samples, int64_t* inds, float* cdf_out, float* z_fine, float* z_std, hipStream_t s, Draws draws) {
    if (n_rays <= 0) return IDN_OK;
    const int nb = S - 1;
    if (S < 3 || S > 64 * kMaxSpl || nb > kMaxBins - 1) return fail(IDN_EUNSUPPORTED, "march: n_samples %d outside [3, %d]", S, kMaxBins);
    if (Ni < 1 || Ni > kMaxNi) return fail(IDN_EUNSUPPORTED, "march: n_importance %d outside [1, %d]", Ni, kMaxNi);
    if (S + Ni > kMaxFine) return fail(IDN_EUNSUPPORTED, "march: n_samples + n_importance > %d", kMaxFine);
    SampleArgs a{z, nullptr, nullptr, nullptr, u, u_per_ray, (long)n_rays, S, Ni, nb, z_samples, inds, cdf_out, z_fine, z_std, draws};
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
    SampleArgs a{z, weights, cdf_in, bins_in, u, u_per_ray, (long)n_rays, S, Ni, nb, z_samples, inds, cdf_out, z_fine, z_std, Draws{0, 0, 0}};
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, s, a);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
