// AudioNet (models/audio_net.py:43-69 upstream) forward and backward as ONE kernel each (gfx950).
//
// The reference runs it once per frame / training step on ONE DeepSpeech window [16, 29] (audio_exp_nerf.py:258-259; eight
// windows under the attention smoother, :235-257): four kernel-3 stride-2 convolutions (29 -> 32 -> 32 -> 64 -> 64 channels,
// length 16 -> 8 -> 4 -> 2 -> 1), LeakyReLU(0.02) after each, then Linear(64, 64) + LeakyReLU + Linear(64, dim_aud).  That is
// 60 k MACs -- and, as eager ops, ~20 launches forward and ~30 backward, each a few microseconds of GPU time behind a dispatch
// gap: a third of the small launches of a training step (profiles/r04_train_small_ops.log).  Here a workgroup walks the six
// layers with the activations in LDS; the backward is one workgroup for all windows (deterministic: no atomics).
#include "idn_internal.h"

namespace idn {

namespace {
constexpr int kAudL = 16, kAudC = 29;                       // window: 16 frames x 29 DeepSpeech logits
// channels 29 -> 32 -> 32 -> 64 -> 64, lengths 16 -> 8 -> 4 -> 2 -> 1
constexpr int kOff1 = 0, kOff2 = 256, kOff3 = 384, kOff4 = 512, kOffH = 576, kSaved = 640;   // saved activations per window
constexpr float kSlope = 0.02f;

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : kSlope * x; }
__device__ __forceinline__ float dleaky(float post) { return post > 0.f ? 1.f : kSlope; }   // sign(post) == sign(pre)

// out[co][lo] = leaky(b[co] + sum_{ci, k} w[co][ci][k] * in[ci][2 lo + k - 1])   (padding 1, stride 2)
template <int CI, int CO, int LI>
__device__ __forceinline__ void conv_layer(const float* __restrict__ w, const float* __restrict__ b, const float* in, float* out) {
    constexpr int LO = LI / 2;
    for (int e = threadIdx.x; e < CO * LO; e += blockDim.x) {
        const int co = e / LO, lo = e % LO;
        float acc = b[co];
        const float* wr = w + (long)co * CI * 3;
        for (int ci = 0; ci < CI; ++ci) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int li = 2 * lo + k - 1;
                if (li >= 0 && li < LI) acc = __builtin_fmaf(wr[ci * 3 + k], in[ci * LI + li], acc);
            }
        }
        out[e] = leaky(acc);
    }
}
__device__ __forceinline__ void fc_layer(const float* __restrict__ w, const float* __restrict__ b, const float* in, float* out, int n_in,
                                         int n_out, bool act) {
    for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
        float acc = b[o];
        for (int i = 0; i < n_in; ++i) acc = __builtin_fmaf(w[(long)o * n_in + i], in[i], acc);
        out[o] = act ? leaky(acc) : acc;
    }
}
}  // namespace

__global__ __launch_bounds__(256) void audio_net_fwd_kernel(idn_audio_net_params p, const float* windows, float* out, float* saved) {
    __shared__ float x0[kAudC * kAudL], a[kSaved];
    const int n = blockIdx.x;
    const float* win = windows + (long)n * kAudL * kAudC;
    for (int e = threadIdx.x; e < kAudC * kAudL; e += blockDim.x) {   // x[c][l] = window[l][c]   (the permute of :63)
        const int l = e / kAudC, c = e % kAudC;
        x0[c * kAudL + l] = win[e];
    }
    __syncthreads();
    conv_layer<29, 32, 16>(p.conv_w[0], p.conv_b[0], x0, a + kOff1);
    __syncthreads();
    conv_layer<32, 32, 8>(p.conv_w[1], p.conv_b[1], a + kOff1, a + kOff2);
    __syncthreads();
    conv_layer<32, 64, 4>(p.conv_w[2], p.conv_b[2], a + kOff2, a + kOff3);
    __syncthreads();
    conv_layer<64, 64, 2>(p.conv_w[3], p.conv_b[3], a + kOff3, a + kOff4);
    __syncthreads();
    fc_layer(p.fc_w[0], p.fc_b[0], a + kOff4, a + kOffH, 64, 64, true);
    __syncthreads();
    fc_layer(p.fc_w[1], p.fc_b[1], a + kOffH, out + (long)n * p.dim_aud, 64, p.dim_aud, false);
    if (saved)
        for (int e = threadIdx.x; e < kSaved; e += blockDim.x) saved[(long)n * kSaved + e] = a[e];
}

// delta of a convolution's INPUT from the delta of its output: din[ci][li] = leaky'(in_post[ci][li]) * sum_{co, (lo, k): 2 lo + k - 1 = li} w[co][ci][k] dout[co][lo]
template <int CI, int CO, int LI>
__device__ __forceinline__ void conv_dinput(const float* __restrict__ w, const float* dout, const float* in_post, float* din) {
    constexpr int LO = LI / 2;
    for (int e = threadIdx.x; e < CI * LI; e += blockDim.x) {
        const int ci = e / LI, li = e % LI;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = li + 1 - k;          // = 2 lo
            if (t < 0 || (t & 1) || t / 2 >= LO) continue;
            const int lo = t / 2;
            for (int co = 0; co < CO; ++co) acc = __builtin_fmaf(w[((long)co * CI + ci) * 3 + k], dout[co * LO + lo], acc);
        }
        din[e] = acc * dleaky(in_post[e]);
    }
}
// dW[co][ci][k] = sum_n sum_lo dout[n][co][lo] * in[n][ci][2 lo + k - 1];  db[co] = sum_n sum_lo dout[n][co][lo]
template <int CI, int CO, int LI>
__device__ __forceinline__ void conv_dweight(const float* dout, int dout_stride, const float* in, int in_stride, int n, float* dW, float* db) {
    constexpr int LO = LI / 2;
    for (int e = threadIdx.x; e < CO * CI * 3; e += blockDim.x) {
        const int co = e / (CI * 3), ci = (e / 3) % CI, k = e % 3;
        float acc = 0.f;
        for (int w = 0; w < n; ++w)
            for (int lo = 0; lo < LO; ++lo) {
                const int li = 2 * lo + k - 1;
                if (li >= 0 && li < LI) acc = __builtin_fmaf(dout[w * dout_stride + co * LO + lo], in[w * in_stride + ci * LI + li], acc);
            }
        dW[e] = acc;
    }
    for (int co = threadIdx.x; co < CO; co += blockDim.x) {
        float acc = 0.f;
        for (int w = 0; w < n; ++w)
            for (int lo = 0; lo < LO; ++lo) acc += dout[w * dout_stride + co * LO + lo];
        db[co] = acc;
    }
}

// One workgroup, all n <= kAudMaxBwd windows.  LDS per window: the input (464), the saved activations (640), the deltas of
// the five hidden outputs (640) and of the output (dim_aud <= 128).
constexpr int kAudMaxBwd = 8, kAudMaxDim = 128;
constexpr int kWinFloats = kAudC * kAudL + 2 * kSaved + kAudMaxDim;   // 1872
__global__ __launch_bounds__(1024) void audio_net_bwd_kernel(idn_audio_net_params p, idn_audio_net_grads g, const float* windows,
                                                             const float* saved, const float* d_out, int n) {
    extern __shared__ float sm[];
    const int D = p.dim_aud;
    auto X = [&](int w) { return sm + w * kWinFloats; };                      // x[c][l]
    auto A = [&](int w) { return sm + w * kWinFloats + kAudC * kAudL; };      // saved post-activations
    auto Dl = [&](int w) { return sm + w * kWinFloats + kAudC * kAudL + kSaved; };          // deltas, same offsets as A
    auto Do = [&](int w) { return sm + w * kWinFloats + kAudC * kAudL + 2 * kSaved; };      // d out
    for (int w = 0; w < n; ++w) {
        for (int e = threadIdx.x; e < kAudC * kAudL; e += blockDim.x) X(w)[(e % kAudC) * kAudL + e / kAudC] = windows[(long)w * kAudC * kAudL + e];
        for (int e = threadIdx.x; e < kSaved; e += blockDim.x) A(w)[e] = saved[(long)w * kSaved + e];
        for (int e = threadIdx.x; e < D; e += blockDim.x) Do(w)[e] = d_out[(long)w * D + e];
    }
    __syncthreads();
    // the delta chain, window by window
    for (int w = 0; w < n; ++w) {
        for (int i = threadIdx.x; i < 64; i += blockDim.x) {            // d h = (W6^T d out) . leaky'(h)
            float acc = 0.f;
            for (int o = 0; o < D; ++o) acc = __builtin_fmaf(p.fc_w[1][(long)o * 64 + i], Do(w)[o], acc);
            Dl(w)[kOffH + i] = acc * dleaky(A(w)[kOffH + i]);
        }
    }
    __syncthreads();
    for (int w = 0; w < n; ++w) {
        for (int i = threadIdx.x; i < 64; i += blockDim.x) {            // d a4 = (W5^T d h) . leaky'(a4)
            float acc = 0.f;
            for (int o = 0; o < 64; ++o) acc = __builtin_fmaf(p.fc_w[0][(long)o * 64 + i], Dl(w)[kOffH + o], acc);
            Dl(w)[kOff4 + i] = acc * dleaky(A(w)[kOff4 + i]);
        }
    }
    __syncthreads();
    for (int w = 0; w < n; ++w) conv_dinput<64, 64, 2>(p.conv_w[3], Dl(w) + kOff4, A(w) + kOff3, Dl(w) + kOff3);
    __syncthreads();
    for (int w = 0; w < n; ++w) conv_dinput<32, 64, 4>(p.conv_w[2], Dl(w) + kOff3, A(w) + kOff2, Dl(w) + kOff2);
    __syncthreads();
    for (int w = 0; w < n; ++w) conv_dinput<32, 32, 8>(p.conv_w[1], Dl(w) + kOff2, A(w) + kOff1, Dl(w) + kOff1);
    __syncthreads();
    // weight and bias gradients, summed over the windows in a fixed order
    conv_dweight<29, 32, 16>(Dl(0) + kOff1, kWinFloats, X(0), kWinFloats, n, g.conv_w[0], g.conv_b[0]);
    conv_dweight<32, 32, 8>(Dl(0) + kOff2, kWinFloats, A(0) + kOff1, kWinFloats, n, g.conv_w[1], g.conv_b[1]);
    conv_dweight<32, 64, 4>(Dl(0) + kOff3, kWinFloats, A(0) + kOff2, kWinFloats, n, g.conv_w[2], g.conv_b[2]);
    conv_dweight<64, 64, 2>(Dl(0) + kOff4, kWinFloats, A(0) + kOff3, kWinFloats, n, g.conv_w[3], g.conv_b[3]);
    for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {            // Linear(64, 64): dW5[o][i] = sum_w d h[o] a4[i]
        const int o = e / 64, i = e % 64;
        float acc = 0.f;
        for (int w = 0; w < n; ++w) acc = __builtin_fmaf(Dl(w)[kOffH + o], A(w)[kOff4 + i], acc);
        g.fc_w[0][e] = acc;
    }
    for (int o = threadIdx.x; o < 64; o += blockDim.x) {
        float acc = 0.f;
        for (int w = 0; w < n; ++w) acc += Dl(w)[kOffH + o];
        g.fc_b[0][o] = acc;
    }
    for (int e = threadIdx.x; e < D * 64; e += blockDim.x) {             // Linear(64, dim_aud)
        const int o = e / 64, i = e % 64;
        float acc = 0.f;
        for (int w = 0; w < n; ++w) acc = __builtin_fmaf(Do(w)[o], A(w)[kOffH + i], acc);
        g.fc_w[1][e] = acc;
    }
    for (int o = threadIdx.x; o < D; o += blockDim.x) {
        float acc = 0.f;
        for (int w = 0; w < n; ++w) acc += Do(w)[o];
        g.fc_b[1][o] = acc;
    }
}

size_t audio_net_saved_floats(int n) { return (size_t)(n > 0 ? n : 0) * kSaved; }

static int check_params(const idn_audio_net_params* p) {
    if (!p) return fail(IDN_EINVAL, "audio_net: params is NULL");
    for (int i = 0; i < 4; ++i)
        if (!p->conv_w[i] || !p->conv_b[i]) return fail(IDN_EINVAL, "audio_net: NULL convolution parameter %d", i);
    for (int i = 0; i < 2; ++i)
        if (!p->fc_w[i] || !p->fc_b[i]) return fail(IDN_EINVAL, "audio_net: NULL linear parameter %d", i);
    if (p->dim_aud < 1 || p->dim_aud > kAudMaxDim) return fail(IDN_EUNSUPPORTED, "audio_net: dim_aud %d outside [1, %d]", p->dim_aud, kAudMaxDim);
    return IDN_OK;
}

int launch_audio_net_fwd(const idn_audio_net_params* p, const float* windows, int n, float* out, float* saved, hipStream_t s) {
    if (int e = check_params(p)) return e;
    if (n < 0) return fail(IDN_EINVAL, "audio_net: n < 0");
    if (n == 0) return IDN_OK;
    if (!windows || !out) return fail(IDN_EINVAL, "audio_net: NULL pointer");
    hipLaunchKernelGGL(audio_net_fwd_kernel, dim3(n), dim3(256), 0, s, *p, windows, out, saved);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

int launch_audio_net_bwd(const idn_audio_net_params* p, const idn_audio_net_grads* g, const float* windows, const float* saved,
                         const float* d_out, int n, hipStream_t s) {
    if (int e = check_params(p)) return e;
    if (!g || !windows || !saved || !d_out) return fail(IDN_EINVAL, "audio_net_bwd: NULL pointer");
    for (int i = 0; i < 4; ++i)
        if (!g->conv_w[i] || !g->conv_b[i]) return fail(IDN_EINVAL, "audio_net_bwd: NULL gradient buffer");
    for (int i = 0; i < 2; ++i)
        if (!g->fc_w[i] || !g->fc_b[i]) return fail(IDN_EINVAL, "audio_net_bwd: NULL gradient buffer");
    if (n < 1 || n > kAudMaxBwd) return fail(IDN_EUNSUPPORTED, "audio_net_bwd: %d windows (built for 1 .. %d: a frame's window, or the smoother's eight)", n, kAudMaxBwd);
    const size_t lds = (size_t)n * kWinFloats * sizeof(float);
    static LaunchSetup setup;
    int num_cu = 0;
    if (int e = setup.get([]() -> int {
            IDN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&audio_net_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)(kAudMaxBwd * kWinFloats * sizeof(float))));
            return IDN_OK;
        }, &num_cu))
        return e;
    hipLaunchKernelGGL(audio_net_bwd_kernel, dim3(1), dim3(1024), lds, s, *p, *g, windows, saved, d_out, n);
    IDN_HIP_CHECK(hipGetLastError());
    return IDN_OK;
}

}  // namespace idn
