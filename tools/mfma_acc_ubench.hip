// Microbenchmark: v_mfma_f32_32x32x2_f32 rate as a function of how the accumulators are visited --
// the dW GEMM of the training step keeps a 4x4 grid of 32x32 accumulator tiles per wave (256 AGPRs).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_acc_ubench.hip -o /tmp/ub_acc
//   VAR 0: one accumulator, dependent chain                      (the MLP kernels' pattern)
//   VAR 1: 16 accumulators, round-robin: acc[x][y] += a[x] b[y]  (outer product per point pair)
//   VAR 2: 16 accumulators, each visited 8 times in a row        (8 point pairs per accumulator)
//   VAR 3: 4 accumulators round-robin
//   VAR 4: 16 accumulators, each visited 2 times in a row
//   VAR 5: 16 accumulators, each visited 4 times in a row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4][4];
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y)
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    float a[8][4], b[8][4];
    for (int s = 0; s < 8; ++s)
        for (int j = 0; j < 4; ++j) {
            a[s][j] = out[lane + 64 * ((s * 4 + j) & 3)] + 0.001f * (s * 4 + j);
            b[s][j] = out[lane + 64 * ((s + j) & 3)] - 0.002f * (s * 4 + j);
        }
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {   // 128 MFMAs per iteration in every variant
        if (VAR == 0) {
#pragma unroll
            for (int u = 0; u < 128; ++u) MF(acc[0][0], a[u & 7][u & 3], b[(u >> 3) & 7][(u >> 1) & 3]);
        } else if (VAR == 1) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) MF(acc[x][y], a[s][x], b[s][y]);
        } else if (VAR == 2) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int s = 0; s < 8; ++s) MF(acc[x][y], a[s][x], b[s][y]);
        } else if (VAR == 3) {
#pragma unroll
            for (int s = 0; s < 32; ++s)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) MF(acc[x][y], a[s & 7][x + 2 * (s >> 4)], b[s & 7][y + 2 * ((s >> 3) & 1)]);
        } else if (VAR == 4 || VAR == 5) {
            constexpr int RUN = VAR == 4 ? 2 : 4;
#pragma unroll
            for (int s0 = 0; s0 < 8; s0 += RUN)
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y)
#pragma unroll
                        for (int s = s0; s < s0 + RUN; ++s) MF(acc[x][y], a[s][x], b[s][y]);
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y)
            for (int r = 0; r < 16; ++r) s += acc[x][y][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The dW GEMM's chunk loop, piece by piece: 16 accumulators, operands from LDS by pipelined asm reads
//   LVAR 0: LDS reads only                      LVAR 1: + a raw barrier per 8 point pairs
//   LVAR 2: + one LDS-DMA piece per point pair  LVAR 3: + both
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int LVAR>
__global__ __launch_bounds__(256, 1) void kl(float* out, const float* src, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 3 x 16 x 512 floats
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 3 * 16 * 512; i += 256) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    f32x16 acc[4][4];
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y)
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    const uint32_t a_addr0 = base + ((lane >> 5) * 256 + 4 * (lane & 31)) * 4, b_addr0 = a_addr0 + 16 * 256 * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0xfffffffc, 0x00020000);
    const uint32_t voff = lane * 16;
    auto mm = [&](const f32x4& a, const f32x4& b) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) MF(acc[x][y], a[x], b[y]);
    };
    const unsigned long long t0 = clock64();
    int buf = 0;
    uint32_t soff = blockIdx.x * 65536u;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {   // one chunk: 8 point pairs = 128 MFMAs
        if (LVAR & 1) {
            if (LVAR & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        uint32_t pa = a_addr0 + buf * (16 * 512 * 4), pb = b_addr0 + buf * (16 * 512 * 4);
        const int buf2 = buf >= 1 ? buf - 1 : 2;
        f32x4 a0, b0, a1, b1;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a0) : "v"(pa) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(b0) : "v"(pb) : "memory");
#pragma unroll 1
        for (int s2 = 0; s2 < 4; ++s2) {
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a1) : "v"(pa) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b1) : "v"(pb) : "memory");
            if (LVAR & 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + buf2 * 16 * 512 + (4 * w + 2 * s2) * 256), 16, voff, soff, 0, 0);
                soff += 1024;
            }
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0), "+v"(b0)::"memory");
            mm(a0, b0);
            const bool last = s2 == 3;
            pa = last ? pa : pa + 4096;
            pb = last ? pb : pb + 4096;
            asm volatile("ds_read_b128 %0, %1" : "=v"(a0) : "v"(pa) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(b0) : "v"(pb) : "memory");
            if (LVAR & 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + buf2 * 16 * 512 + (4 * w + 2 * s2 + 1) * 256), 16, voff, soff, 0, 0);
                soff += 1024;
            }
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a1), "+v"(b1)::"memory");
            mm(a1, b1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(b0)::"memory");
        buf = buf == 2 ? 0 : buf + 1;
        if ((it & 63) == 63) soff = blockIdx.x * 65536u;   // stay inside the source buffer
    }
    const unsigned long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y)
            for (int r = 0; r < 16; ++r) s += acc[x][y][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int LVAR>
void runl(const char* name, float* out, const float* src, unsigned long long* cyc, int blocks) {
    const int iters = 1000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&kl<LVAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kl<LVAR>, dim3(blocks), dim3(256), 3 * 16 * 512 * 4, 0, out, src, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kl<LVAR>, dim3(blocks), dim3(256), 3 * 16 * 512 * 4, 0, out, src, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double n = (double)iters * 128;
    printf("%-52s %7.2f cycles/MFMA   %7.1f TFLOP/s (wall, %d CUs)\n", name, avg / n, n * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12, blocks);
}

template <int VAR>
void run(const char* name, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 1000;
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
    const double n = (double)iters * 128;
    printf("%-52s %7.2f cycles/MFMA   %7.1f TFLOP/s (wall, %d CUs)\n", name, avg / n, n * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12, blocks);
}

int main() {
    int blocks = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4 + 1024); hipMemset(out, 0, blocks * 256 * 4 + 1024);
    hipMalloc(&cyc, blocks * 8);
    run<0>("1 accumulator, dependent chain", out, cyc, blocks);
    run<1>("16 accumulators, round-robin (outer product)", out, cyc, blocks);
    run<2>("16 accumulators, 8 in a row on each", out, cyc, blocks);
    run<3>("4 accumulators, round-robin", out, cyc, blocks);
    run<4>("16 accumulators, 2 in a row on each", out, cyc, blocks);
    run<5>("16 accumulators, 4 in a row on each", out, cyc, blocks);
    float* src;
    hipMalloc(&src, (size_t)blocks * 65536 * 4 + (1 << 20));
    hipMemset(src, 0, (size_t)blocks * 65536 * 4 + (1 << 20));
    runl<0>("GEMM loop: pipelined LDS operand reads", out, src, cyc, blocks);
    runl<1>("GEMM loop: + barrier per chunk", out, src, cyc, blocks);
    runl<2>("GEMM loop: + one LDS-DMA piece per point pair", out, src, cyc, blocks);
    runl<3>("GEMM loop: + barrier + pieces", out, src, cyc, blocks);
    return 0;
}
