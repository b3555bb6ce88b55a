// Microbenchmark: what HBM takes when every CU writes rows of a [P, 256] fp32 matrix the way the bf16x6 training
// kernels record a layer's activations ("Training kernels"):
//   VAR 0: a lane owns one row (point): four dwordx4 stores per 32-channel tile, each instruction touching 32 rows with
//          32 bytes (two lanes) per row -- the accumulator layout's natural store
//   VAR 1: the same bytes into the same 32 rows, 1 KiB contiguous per instruction (whole 128-byte lines)
//   VAR 2: VAR 0 with a long stretch of dependent arithmetic between the bursts of 32 stores (a layer's MFMA phase)
//   VAR 3: that arithmetic alone (no stores): VAR 2 - VAR 3 = what the bursts cost when nothing waits for them
//   VAR 4: VAR 0 with non-temporal stores (`global_store ... nt`)
// hipcc --offload-arch=gfx950 -O3 tools/hbm_write_ubench.hip -o /tmp/ub_wr && /tmp/ub_wr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, long p_pad, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const long ntiles = p_pad >> 7;
    float acc = (float)lane;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p0 = tile * 128 + wave * 32;
        for (int l = 0; l < layers; ++l) {
            float* mat = out + (long)l * p_pad * 256;
            if (VAR == 2 || VAR == 3) {
#pragma unroll 1
                for (int i = 0; i < 3000; ++i) acc = acc * 1.0001f + 0.5f;   // ~12 000 cycles of dependent VALU
            }
            const f32x4 v = {acc, acc + 1.f, acc + 2.f, acc + 3.f};
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (VAR == 1) *reinterpret_cast<f32x4*>(mat + p0 * 256 + (t * 4 + q) * 256 + lane * 4) = v;
                    else if (VAR == 3) acc += v.x * 1e-30f;
                    else if (VAR == 4) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(mat + (p0 + m) * 256 + 32 * t + 8 * q + 4 * h));
                    else *reinterpret_cast<f32x4*>(mat + (p0 + m) * 256 + 32 * t + 8 * q + 4 * h) = v;
                }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int VAR>
static void run(float* buf, long p_pad, int layers, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(256), 0, 0, buf, p_pad, layers);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(256), 0, 0, buf, p_pad, layers);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double gb = (double)p_pad * 256 * 4 * layers / 1e9;
    printf("%-78s %.3f ms per launch, %.2f GB -> %.2f TB/s\n", what, ms / 5, gb, gb / (ms / 5));
}

int main() {
    const long p_pad = 589824;   // the fine pass of the train step: 3072 rays x 192 points
    const int layers = 8;        // 8 x 256 columns = 4.8 GB per launch
    float* buf;
    if (hipMalloc(&buf, (size_t)p_pad * 256 * 4 * layers) != hipSuccess) return 1;
    run<0>(buf, p_pad, layers, "lane = row, dwordx4 (32 rows x 32 B per instruction):");
    run<1>(buf, p_pad, layers, "same bytes, 1 KiB contiguous per instruction:");
    run<4>(buf, p_pad, layers, "lane = row, dwordx4, non-temporal:");
    run<2>(buf, p_pad, layers, "lane = row, bursts of 32 stores with dependent arithmetic in between:");
    run<3>(buf, p_pad, layers, "that arithmetic alone:");
    run<0>(buf, p_pad, layers, "lane = row, dwordx4 (again):");
    hipFree(buf);
    return 0;
}
