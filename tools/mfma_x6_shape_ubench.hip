// The six-piece (bf16x6) step under the two bf16 MFMA shapes, with the kernel's own operand reuse: per step three weight
// fragments from LDS (p1, p2, p3) and six piece products per 32 points, B pieces constant over the eight tile-steps of a
// k-step, ~NV vector instructions per step (the conversions in the MFMA shadow), random data, seconds of load.
//   32x32x16: 6 MFMAs of 32 cycles on one 32 x 32 accumulator tile          [what mlp_bf16x6.hip issues]
//   16x16x32: 12 MFMAs of 16 cycles on two 16 x 16 tiles (two 16-point groups share the A fragments)
// Equal FLOPs, equal LDS bytes, equal VALU per step.   hipcc --offload-arch=gfx950 -O3 tools/mfma_x6_shape_ubench.hip -o tools/mfma_x6_shape_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define BC(x) __builtin_bit_cast(bf16x8, x)

template <int SHAPE, int NV>
__global__ __launch_bounds__(256, 1) void k(const float* init, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 8 * 256 * 2];  // 48 KiB: two k-steps of (8 tiles x 3 fragments)
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3 * 8 * 256 * 2; i += 256) lds[i] = init[i & 16383];
    __syncthreads();
    f32x16 acc32[8];
    f32x4 acc16[16];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc32[t][r] = 0.f;
    for (int t = 0; t < 16; ++t) acc16[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b[2][3];   // the pieces of two 16-point groups (32x32x16 uses group 0's as its 32-point operand)
    for (int g = 0; g < 2; ++g) for (int q = 0; q < 3; ++q) b[g][q] = *reinterpret_cast<const f32x4*>(init + 1024 * (1 + 3 * g + q) + lane * 4);
    float side[4] = {init[lane], init[lane + 1], init[lane + 2], init[lane + 3]};
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds + lane * 16;
    f32x4 fa[2][3];   // two fragment sets, ping-pong: step u multiplies set u & 1 while set (u + 1) & 1 is being read
    for (int q = 0; q < 3; ++q) fa[0][q] = b[0][q];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {   // two k-steps of eight tile-steps
            const int t = u & 7;
            f32x4 (&a)[3] = fa[u & 1];
            f32x4 (&an)[3] = fa[(u + 1) & 1];
#pragma unroll
            for (int q = 0; q < 3; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(an[q]) : "v"(addr), "n"(0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])::"memory");
            if constexpr (SHAPE == 32) {
#define T6(QA, QB) acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BC(a[QA]), BC(b[0][QB]), acc32[t], 0, 0, 0);
                T6(0, 0) T6(0, 1) T6(1, 0)
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_max_i32 %0, %0, %1" : "+v"(side[v & 3]) : "v"(b[0][0].x));
                T6(1, 1) T6(0, 2) T6(2, 0)
#undef T6
            } else {
#define T12(QA, QB) acc16[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BC(a[QA]), BC(b[0][QB]), acc16[2 * t], 0, 0, 0); \
                    acc16[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BC(a[QA]), BC(b[1][QB]), acc16[2 * t + 1], 0, 0, 0);
                T12(0, 0) T12(0, 1) T12(1, 0)
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_max_i32 %0, %0, %1" : "+v"(side[v & 3]) : "v"(b[0][0].x));
                T12(1, 1) T12(0, 2) T12(2, 0)
#undef T12
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t == 7) {   // a new k-step: new pieces (a cheap permutation of the old ones keeps the data random)
                for (int g = 0; g < 2; ++g) { f32x4 tmp = b[g][0]; b[g][0] = b[g][1]; b[g][1] = b[g][2]; b[g][2] = tmp; }
            }
        }
        if ((it & 63) == 63) {
            for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc32[t][r] *= 1e-3f;
            for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) acc16[t][r] *= 1e-3f;
        }
    }
    float s = side[0] + side[1] + side[2] + side[3];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc32[t][r];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc16[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NV>
void run(const char* name, const float* init, float* out, int blocks, double seconds) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 7000;   // 7000 x 16 steps x 192 cycles = ~10 ms per launch
    float ms = 0, total = 0;
    int n = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, NV>), dim3(blocks), dim3(256), 0, 0, init, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        total += ms; ++n;
    }
    const double flop = (double)iters * 16 * 6 * 32768.0 * 4 * blocks;
    printf("%-52s last launch %7.3f ms  %7.1f TFLOP/s of MFMAs = %.3f of 2516.6  (%d launches)\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2516.6, n);
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
    run<32, 12>("32x32x16: 3 frags + 6 MFMAs + 12 VALU per step", init, out, blocks, secs);
    run<16, 12>("16x16x32: 3 frags + 12 MFMAs + 12 VALU per step", init, out, blocks, secs);
    run<32, 0>("32x32x16: 3 frags + 6 MFMAs, no VALU", init, out, blocks, secs);
    run<16, 0>("16x16x32: 3 frags + 12 MFMAs, no VALU", init, out, blocks, secs);
    run<32, 12>("32x32x16: 3 frags + 6 MFMAs + 12 VALU per step", init, out, blocks, secs);
    run<16, 12>("16x16x32: 3 frags + 12 MFMAs + 12 VALU per step", init, out, blocks, secs);
    return 0;
}
