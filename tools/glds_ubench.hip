// Microbenchmark: what does one LDS-DMA piece (1 KiB per wave) cost the issuing wave beside a
// v_mfma_f32_32x32x16_bf16 chain, by addressing form?  One wave per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/glds_ubench.hip -o tools/glds_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int kSrcBytes = 2304 * 1024;  // the bf16x3 weight stream's size

// KIND 0 none | 1 global_load_lds x4, 64-bit per-lane pointer | 2 raw_buffer_load_lds x4, offen (32-bit voffset)
//      3 raw_buffer_load_lds x4, no VGPR (ADD_TID_ENABLE descriptor) | 4 global_load_dwordx4 into VGPRs (no LDS)
//      5 global_load_lds x4 saddr form (uniform base + 32-bit lane offset)
template <int KIND, int NM, int NR = 0, int NV = 0, int PLACE = 0>
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc = {0};
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {out[lane], out[lane + 64], out[lane + 128], out[lane + 192]};
    char* ring_wave = lds + wave * 1024;
    const char* gl = src + tid * 16;
    unsigned pos = 0;  // wave-uniform byte position of the piece in the stream
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, -1, 0x00020000);  // raw, untyped
    __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)src, 16, -1, (1 << 23));  // ADD_TID_ENABLE, stride 16 (word3[18:15] are stride[17:14] in this mode: keep 0)
    f32x4 sink = {0, 0, 0, 0}, a2 = a;
    const uint32_t raddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
    float side[8];
    for (int i = 0; i < 8; ++i) side[i] = out[lane + i];
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            char* dst = ring_wave + ((it * 8 + u) & 15) * 4096;
            auto piece = [&]() {
                if constexpr (KIND == 1) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gl + pos), LDS_PTR(dst), 16, 0, 0);
                if constexpr (KIND == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, tid * 16, pos, 0, 0);
                if constexpr (KIND == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, LDS_PTR(dst), 16, 0, pos + wave * 1024, 0, 0);
                if constexpr (KIND == 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(gl + pos);
                asm volatile("" : "+v"(v));
                sink = v;
                }
                if constexpr (KIND == 5) {
                const char* sb = src + pos;  // uniform
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sb + (unsigned)(tid * 16)), LDS_PTR(dst), 16, 0, 0);
                }
            };
            if constexpr (PLACE == 0) piece();
            pos += 4096;
            if (pos >= kSrcBytes) pos = 0;
            // NR fragment reads per step, issued one step ahead of their use (as the MLP kernel does)
            f32x4 n0 = a, n1 = a2;
            if constexpr (NR >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(n0) : "v"(raddr), "n"(1024 * (u & 7)) : "memory");
            if constexpr (NR >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(n1) : "v"(raddr), "n"(1024 * (u & 7) + 8192) : "memory");
            if constexpr (NR >= 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(a2) : "n"(NR) : "memory");
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc = mf(m & 1 ? a2 : a, b, acc);
                if (PLACE == m + 1) piece();
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(side[v & 7]) : "v"(b.x));
            }
            a = n0;
            a2 = n1;
            if constexpr (KIND != 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = sink.x + lds[tid] + a2.x;
    for (int i = 0; i < 8; ++i) s += side[i];
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NM, int NR = 0, int NV = 0, int PLACE = 0>
void run(const char* name, const char* src, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, NM, NR, NV, PLACE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<KIND, NM, NR, NV, PLACE>), dim3(blocks), dim3(256), 65536, 0, src, out, cyc, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NM, NR, NV, PLACE>), dim3(blocks), dim3(256), 65536, 0, src, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double n = (double)iters * 8;
    printf("%-58s %7.1f ticks/step  %7.2f ns/step (%d MFMA/step => floor %d)  err=%s\n", name, avg / n, ms * 1e6 / n, NM, NM * 32,
           hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    int blocks = 256;
    float* out; unsigned long long* cyc; char* src;
    (void)hipMalloc(&out, blocks * 256 * 4 + 1024); (void)hipMemset(out, 0, blocks * 256 * 4 + 1024);
    (void)hipMalloc(&cyc, blocks * 8);
    (void)hipMalloc(&src, kSrcBytes + 8192); (void)hipMemset(src, 0, kSrcBytes + 8192);
    run<0, 3, 2, 4>("3 MFMA + 2 ds_read + 12 VALU", src, out, cyc, blocks);
    run<2, 3, 2, 4, 0>("  + buffer piece at the top of the step", src, out, cyc, blocks);
    run<2, 3, 2, 4, 1>("  + buffer piece after MFMA 1", src, out, cyc, blocks);
    run<2, 3, 2, 4, 2>("  + buffer piece after MFMA 2", src, out, cyc, blocks);
    run<2, 3, 2, 4, 3>("  + buffer piece after MFMA 3 (+VALU)", src, out, cyc, blocks);
    run<0, 3, 2, 2>("3 MFMA + 2 ds_read + 6 VALU", src, out, cyc, blocks);
    run<2, 3, 2, 2, 0>("  + buffer piece at the top of the step", src, out, cyc, blocks);
    run<2, 3, 2, 2, 1>("  + buffer piece after MFMA 1", src, out, cyc, blocks);
    run<2, 3, 2, 2, 2>("  + buffer piece after MFMA 2", src, out, cyc, blocks);
    run<0, 2, 2, 3>("2 MFMA + 2 ds_read + 6 VALU", src, out, cyc, blocks);
    run<2, 2, 2, 3, 0>("  + buffer piece at the top of the step", src, out, cyc, blocks);
    run<2, 2, 2, 3, 1>("  + buffer piece after MFMA 1", src, out, cyc, blocks);
    run<2, 2, 2, 3, 2>("  + buffer piece after MFMA 2", src, out, cyc, blocks);
    return 0;
}
