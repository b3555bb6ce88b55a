// The ray march as device functions, one ray per wavefront: raw2outputs (a6), sample_pdf (a7) and the sorted merge (a8).
// Used by the stand-alone kernels of composite.hip and by the fused ray kernel (render_fused.hip).  gfx950 only.
#pragma once
#include "idn_internal.h"

namespace idn {

constexpr int kMaxSpl = 4;  // samples per lane: S <= 256

// Column `col` of row `row` of the draw table (include/idealnerf.h: idealnerf_philox_uniform): Philox4x32-10 with
// key = seed, counter = (col / 4, 0, row); word col % 4 as a 24-bit uniform on [0, 1).
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned long long row, unsigned col) {
    unsigned c0 = col >> 2, c1 = 0u, c2 = (unsigned)row, c3 = (unsigned)(row >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned w = col & 3u;
    const unsigned x = w == 0 ? c0 : w == 1 ? c1 : w == 2 ? c2 : c3;
    return (float)(x >> 8) * 0x1p-24f;
}
__device__ __forceinline__ float draw_t_rand(const Draws& d, long ray, int s) { return philox_uniform(d.seed, 2ull * (unsigned long long)(d.ray0 + ray), (unsigned)s); }
__device__ __forceinline__ float draw_u(const Draws& d, long ray, int j) { return philox_uniform(d.seed, 2ull * (unsigned long long)(d.ray0 + ray) + 1ull, (unsigned)j); }

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
// a6: raw2outputs (NeRFs/HeadNeRF/train/baseline.py:325-375; rgb_fg: TorsoNeRF/run_nerf.py:757)
// Lane l owns samples l*SPL .. l*SPL+SPL-1 (contiguous, so a ray's prefix product is a
// lane-local product followed by one wave scan).
// ---------------------------------------------------------------------------
// One ray per wave.  wout[i] = weight of sample lane * SPL + i (0 beyond S), for a caller that goes on with them.
// rawr / zr: the ray's S raw outputs and depths (global memory, or LDS in the fused ray kernel); the per-ray inputs
// (rays, bc, noise) and all outputs are indexed with `ray`.
template <int SPL>
__device__ __forceinline__ void composite_ray(const float4* rawr, const float* zr, const float* rays, const float* bc,
                                              long ray, int lane, int S, const float* noise, int white_bkgd,
                                              const idn_composite_out& out, float (&wout)[SPL]) {
    const float* rr = rays + ray * IDN_RAY_FLOATS;
    const float dn = sqrtf((rr[3] * rr[3] + rr[4] * rr[4]) + rr[5] * rr[5]);  // torch.norm(rays_d)

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
    Draws draws;           // on: u is drawn here, row draws.ray0 + ray of the table (a.u is not read)
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
// `zfine_lds` (optional): the merged depths also (or only, when a.z_fine is null) go to this LDS row -- the fused ray kernel
// keeps them on chip for the fine pass.
__device__ __forceinline__ void sample_pdf_ray(const SampleArgs& a, long ray, int lane, float* cdf, float* bins, float* val,
                                               const float* w_lds, float* zfine_lds = nullptr) {
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
            const float u = a.draws.on ? draw_u(a.draws, ray, i) : a.u_per_ray ? a.u[ray * a.Ni + i] : a.u[i];
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
    if (a.z_fine || zfine_lds) {
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
                if (a.z_fine) a.z_fine[ray * nf + rank] = v;
                if (zfine_lds) zfine_lds[rank] = v;
            }
        } else {
            for (int e = lane; e < nf; e += 64) {
                const float v = val[e];
                int rank = 0;
                for (int j = 0; j < nf; ++j) {
                    const float o = val[j];
                    rank += (sort_lt(o, v) || (sort_eq(o, v) && j < e)) ? 1 : 0;
                }
                if (a.z_fine) a.z_fine[ray * nf + rank] = v;
                if (zfine_lds) zfine_lds[rank] = v;
            }
        }
    }
}

}  // namespace idn
