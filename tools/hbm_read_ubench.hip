// Microbenchmark: what HBM gives a kernel that READS two [P, 256] fp32 matrices the way the weight-gradient GEMMs of the train
// step do (gemm_tn_x6_kernel / gemm_tn_kernel stream the rows of a delta matrix and of an activation
// matrix, a workgroup per (product, split of the points), 16 rows of 1 KiB per matrix and chunk, a dword per lane and row) --
// with nothing but an add per loaded value, so that the rate is the memory system's for the access pattern:
//   VAR 0: the GEMMs' pattern: WG (item, split) walks its rows in chunks of 16; thread t loads column t of each row (dword)
//   VAR 1: the same rows, 16 bytes per lane (a wave covers a row; 4 instructions per matrix and chunk instead of 16)
//   VAR 2: VAR 0 with two chunks in flight (32 + 32 loads outstanding per thread)
//   VAR 3: VAR 1 with four chunks in flight (16 + 16 dwordx4 loads outstanding per thread)
//   VAR 4: no structure at all: every thread of a full-chip grid strides over the whole buffer with dwordx4 loads
// splits: workgroups per item (56 = the shipped launch: 504 workgroups, two rounds per CU; 28 = one round; 112 = four)
// hipcc --offload-arch=gfx950 -O3 tools/hbm_read_ubench.hip -o /tmp/ub_rd && /tmp/ub_rd
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kItems = 9;          // 256 x 256 products of a pass
constexpr int kRows = 16;          // rows per chunk

template <int VAR>
__global__ __launch_bounds__(256) void k(const float* a, const float* b, long p_pad, int splits, float* sink) {
    const int item = blockIdx.x / splits, split = blockIdx.x % splits;
    const long rows_per_split = p_pad / splits;                 // (p_pad is a multiple of 16 * splits here)
    const float* A = a + (long)item * p_pad * 256 + (long)split * rows_per_split * 256;
    const float* B = b + (long)item * p_pad * 256 + (long)split * rows_per_split * 256;
    const int t = threadIdx.x;
    float acc = 0.f;
    if (VAR == 0) {
        for (long r0 = 0; r0 < rows_per_split; r0 += kRows) {
            float va[kRows], vb[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) va[r] = A[(r0 + r) * 256 + t];
#pragma unroll
            for (int r = 0; r < kRows; ++r) vb[r] = B[(r0 + r) * 256 + t];
#pragma unroll
            for (int r = 0; r < kRows; ++r) acc += va[r] + vb[r];
        }
    } else if (VAR == 1) {
        const int w = t >> 6, l = t & 63;
        for (long r0 = 0; r0 < rows_per_split; r0 += kRows) {
            f32x4 va[kRows / 4], vb[kRows / 4];
#pragma unroll
            for (int r = 0; r < kRows / 4; ++r) va[r] = *reinterpret_cast<const f32x4*>(A + (r0 + 4 * r + w) * 256 + 4 * l);
#pragma unroll
            for (int r = 0; r < kRows / 4; ++r) vb[r] = *reinterpret_cast<const f32x4*>(B + (r0 + 4 * r + w) * 256 + 4 * l);
#pragma unroll
            for (int r = 0; r < kRows / 4; ++r) acc += (va[r].x + va[r].y) + (va[r].z + va[r].w) + (vb[r].x + vb[r].y) + (vb[r].z + vb[r].w);
        }
    } else if (VAR == 2) {
        for (long r0 = 0; r0 < rows_per_split; r0 += 2 * kRows) {
            float va[2 * kRows], vb[2 * kRows];
#pragma unroll
            for (int r = 0; r < 2 * kRows; ++r) va[r] = A[(r0 + r) * 256 + t];
#pragma unroll
            for (int r = 0; r < 2 * kRows; ++r) vb[r] = B[(r0 + r) * 256 + t];
#pragma unroll
            for (int r = 0; r < 2 * kRows; ++r) acc += va[r] + vb[r];
        }
    } else if (VAR == 3) {
        const int w = t >> 6, l = t & 63;
        for (long r0 = 0; r0 < rows_per_split; r0 += 4 * kRows) {
            f32x4 va[kRows], vb[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) va[r] = *reinterpret_cast<const f32x4*>(A + (r0 + 4 * r + w) * 256 + 4 * l);
#pragma unroll
            for (int r = 0; r < kRows; ++r) vb[r] = *reinterpret_cast<const f32x4*>(B + (r0 + 4 * r + w) * 256 + 4 * l);
#pragma unroll
            for (int r = 0; r < kRows; ++r) acc += (va[r].x + va[r].y) + (va[r].z + va[r].w) + (vb[r].x + vb[r].y) + (vb[r].z + vb[r].w);
        }
    } else {
        const long n4 = (long)kItems * p_pad * 64;   // f32x4 elements per matrix
        const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
        const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
        const long stride = (long)gridDim.x * 256;
        for (long i = (long)blockIdx.x * 256 + t; i < n4; i += 4 * stride) {
            f32x4 v[4], u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (i + j * stride < n4) ? a4[i + j * stride] : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = (i + j * stride < n4) ? b4[i + j * stride] : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w) + (u[j].x + u[j].y) + (u[j].z + u[j].w);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int VAR>
static void run(const float* a, const float* b, long p_pad, int splits, float* sink, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = VAR == 4 ? splits : kItems * splits;
    hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, a, b, p_pad, splits, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, a, b, p_pad, splits, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double gb = 2.0 * kItems * (double)p_pad * 1024 / 1e9;
    printf("%-96s %4d WGs  %.3f ms  %.2f GB -> %.2f TB/s\n", what, grid, ms / 5, gb, gb / (ms / 5));
}

int main() {
    const long p_pad = 602112;   // ~ the fine pass of the train step (589 824 points), a multiple of 16 * 112 * 3 * ... : 16 * 37 632
    float *a, *b, *sink;
    const size_t bytes = (size_t)kItems * p_pad * 1024;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    for (int splits : {28, 56, 112}) {
        printf("-- %d splits per product\n", splits);
        run<0>(a, b, p_pad, splits, sink, "the GEMMs' pattern (dword per lane and row, 16 + 16 rows in flight):");
        run<2>(a, b, p_pad, splits, sink, "the same, 32 + 32 rows in flight:");
        run<1>(a, b, p_pad, splits, sink, "16 bytes per lane (a wave per row), 16 + 16 rows in flight:");
        run<3>(a, b, p_pad, splits, sink, "16 bytes per lane, 64 + 64 rows in flight:");
    }
    printf("-- no structure\n");
    run<4>(a, b, p_pad, 2048, sink, "grid-stride over the whole buffers, dwordx4, 4 + 4 loads in flight per thread:");
    run<4>(a, b, p_pad, 8192, sink, "the same, 8192 workgroups:");
    hipFree(a);
    hipFree(b);
    return 0;
}
