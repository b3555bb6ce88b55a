// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on one wave per SIMD under the operand
// patterns of the MLP kernel.  hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float b0 = out[lane], b1 = out[lane + 64], b2 = out[lane + 128], b3 = out[lane + 192];
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, an = a;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds + lane * 16;
    float side = b0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (VAR == 0) {  // one dependent chain, operands in VGPRs
                MF(acc0, a.x, b0); MF(acc0, a.y, b1); MF(acc0, a.z, b2); MF(acc0, a.w, b3);
            } else if (VAR == 1) {  // four independent accumulators
                MF(acc0, a.x, b0); MF(acc1, a.y, b1); MF(acc2, a.z, b2); MF(acc3, a.w, b3);
            } else if (VAR == 2 || VAR == 3) {  // dependent chain + one asm ds_read_b128 per 4 MFMAs, counted wait
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(an) : "v"(addr), "n"(0) : "memory");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a)::"memory");
                MF(acc0, a.x, b0); MF(acc0, a.y, b1);
                if (VAR == 3) {  // plus a little VALU side work
                    side = fmaxf(side * 1.0001f, 0.f); side = fmaxf(side + b1, 0.f); side = fmaxf(side - b2, 0.f);
                }
                MF(acc0, a.z, b2); MF(acc0, a.w, b3);
                __builtin_amdgcn_sched_barrier(0);
                a = an;
            } else if (VAR == 4) {  // B operand taken from another accumulator tile (AGPR-resident activations)
                MF(acc0, a.x, acc1[0]); MF(acc0, a.y, acc1[1]); MF(acc0, a.z, acc1[2]); MF(acc0, a.w, acc1[3]);
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = side;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR>
void run(const char* name, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double n = (double)iters * 32;
    printf("%-44s %7.2f cycles/MFMA   %7.1f TFLOP/s (wall, %d CUs)\n", name, avg / n, n * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12, blocks);
}

int main() {
    int blocks = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4 + 1024); hipMemset(out, 0, blocks * 256 * 4 + 1024);
    hipMalloc(&cyc, blocks * 8);
    run<0>("dependent chain, VGPR operands", out, cyc, blocks);
    run<1>("4 independent accumulators", out, cyc, blocks);
    run<2>("dependent + ds_read_b128 / 4 MFMA", out, cyc, blocks);
    run<3>("dependent + ds_read + 6 VALU / 4 MFMA", out, cyc, blocks);
    run<4>("dependent, B operand from accumulator regs", out, cyc, blocks);
    return 0;
}
