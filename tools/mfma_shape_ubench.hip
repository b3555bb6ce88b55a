// Which bf16 MFMA shape sustains more FLOP/s under a realistic body (fresh A fragment from LDS for
// every MFMA group, a few VALU per MFMA, random data, seconds of load so DVFS settles)?
//   32x32x16: one ds_read_b128 fragment -> 1 MFMA (32 cycles)      [the layout the MLP kernels use]
//   16x16x32: one ds_read_b128 fragment -> 2 MFMAs (2 x 16 cycles, two 16-point tiles)
// Equal FLOPs and equal LDS bytes per step.  hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define BC(x) __builtin_bit_cast(bf16x8, x)

template <int SHAPE, int NV, int NOLDS = 0>
__global__ __launch_bounds__(256, 1) void k(const float* init, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];  // 64 KiB of random fragments
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = init[i];
    __syncthreads();
    f32x16 acc32 = {0};
    f32x4 acc16a = {0, 0, 0, 0}, acc16b = {0, 0, 0, 0};
    f32x4 b0 = {init[lane], init[64 + lane], init[128 + lane], init[192 + lane]};
    f32x4 b1 = {init[256 + lane], init[320 + lane], init[384 + lane], init[448 + lane]};
    float side[4] = {init[lane], init[lane + 1], init[lane + 2], init[lane + 3]};
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds + lane * 16;
    f32x4 a = b0, an = b0;
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(init + 1024 + lane * 4), r1 = *reinterpret_cast<const f32x4*>(init + 2048 + lane * 4),
                r2 = *reinterpret_cast<const f32x4*>(init + 3072 + lane * 4), r3 = *reinterpret_cast<const f32x4*>(init + 4096 + lane * 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if constexpr (NOLDS) {
                an = (u & 3) == 0 ? r0 : (u & 3) == 1 ? r1 : (u & 3) == 2 ? r2 : r3;  // four register-resident random fragments
            } else {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(an) : "v"(addr), "n"(1024 * (u & 31)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a)::"memory");
            }
            if constexpr (SHAPE == 32) {
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BC(a), BC(u & 1 ? b1 : b0), acc32, 0, 0, 0);
            } else {
                acc16a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BC(a), BC(u & 1 ? b1 : b0), acc16a, 0, 0, 0);
                acc16b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BC(a), BC(u & 1 ? b0 : b1), acc16b, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_max_i32 %0, %0, %1" : "+v"(side[v & 3]) : "v"(b0.x));
            __builtin_amdgcn_sched_barrier(0);
            a = an;
        }
        // keep the accumulators bounded so the data stays "random" instead of saturating
        if ((it & 63) == 63) {
            for (int r = 0; r < 16; ++r) acc32[r] *= 1e-3f;
            for (int r = 0; r < 4; ++r) { acc16a[r] *= 1e-3f; acc16b[r] *= 1e-3f; }
        }
    }
    float s = side[0] + side[1] + side[2] + side[3];
    for (int r = 0; r < 16; ++r) s += acc32[r];
    for (int r = 0; r < 4; ++r) s += acc16a[r] + acc16b[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NV, int NOLDS = 0>
void run(const char* name, const float* init, float* out, int blocks, double seconds) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;  // 20000 x 32 steps x 32 cycles = ~10 ms per launch
    // run back to back for `seconds`, report the LAST launch (steady clock)
    float ms = 0, total = 0;
    int n = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, NV, NOLDS>), dim3(blocks), dim3(256), 0, 0, init, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        total += ms; ++n;
    }
    const double flop = (double)iters * 32 * 32768.0 * 4 * blocks;
    printf("%-40s last launch %7.3f ms  %7.1f TFLOP/s  (%d launches, %.1f s)\n", name, ms, flop / (ms * 1e-3) / 1e12, n, total * 1e-3);
    fflush(stdout);
}

int main() {
    const int blocks = 256;
    std::vector<unsigned> h(16384);
    unsigned st = 777u;
    for (auto& w : h) {
        unsigned v = 0;
        for (int kk = 0; kk < 2; ++kk) { st = st * 1664525u + 1013904223u; unsigned m = (st >> 9) & 0x7f, e = 122 + ((st >> 20) & 7), sg = (st >> 30) & 1; v |= ((sg << 15) | (e << 7) | m) << (16 * kk); }
        w = v;
    }
    float *init, *out;
    (void)hipMalloc(&init, 16384 * 4); (void)hipMemcpy(init, h.data(), 16384 * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, blocks * 256 * 4);
    const double secs = getenv("UB_SECS") ? atof(getenv("UB_SECS")) : 3.0;
    run<32, 2>("32x32x16, frag/MFMA, 2 VALU per 32 cyc", init, out, blocks, secs);
    run<16, 2>("16x16x32, frag/2 MFMA, 2 VALU per 32 cyc", init, out, blocks, secs);
    run<32, 4>("32x32x16, frag/MFMA, 4 VALU per 32 cyc", init, out, blocks, secs);
    run<16, 4>("16x16x32, frag/2 MFMA, 4 VALU per 32 cyc", init, out, blocks, secs);
    run<32, 0>("32x32x16, frag/MFMA, no VALU", init, out, blocks, secs);
    run<16, 0>("16x16x32, frag/2 MFMA, no VALU", init, out, blocks, secs);
    run<32, 0, 1>("32x32x16, 4 register fragments, no LDS", init, out, blocks, secs);
    run<16, 0, 1>("16x16x32, 4 register fragments, no LDS", init, out, blocks, secs);
    return 0;
}
