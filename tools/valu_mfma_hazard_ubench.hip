// How many wait states does an MFMA need before it may read (as its B operand) a VGPR that a vector instruction wrote?
// hipcc's hazard recogniser inserts them for instructions it can see; an inline-asm v_cvt_pk_bf16_f32 it cannot see, and a
// kernel that read such a result two back-to-back MFMAs later produced garbage (round 3).  One asm block per spacing:
//   v_cvt_pk_bf16_f32 b0, x, y ; [s_nop n] ; v_mfma_f32_32x32x16_bf16 acc, a, b[0:3], 0        against the same with a long wait.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_mfma_hazard_ubench.hip -o tools/valu_mfma_hazard_ubench && tools/valu_mfma_hazard_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NOPS, int FILL>   // FILL: independent MFMAs on another accumulator between the write and the read
__global__ void k(const float* in, float* out) {
    const int l = threadIdx.x;
    f32x4 a = {in[l], in[l + 64], in[l + 128], in[l + 192]};
    f32x4 b = {in[l + 256], in[l + 320], in[l + 384], in[l + 448]};
    float x = in[l + 512], y = in[l + 576];
    f32x16 acc, other;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; other[i] = 0.f; }
    // make sure everything has landed and the pipes are idle
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(x), "+v"(y), "+v"(acc), "+v"(other));
    // b lives in v[100:103] (named registers: the conversion writes its FIRST word), copied there well ahead
#define PRE "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\ts_nop 7\n\ts_nop 7\n\t"
#define OPS : "+v"(acc) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(a), "v"(x), "v"(y), "v"(other)
    if constexpr (FILL == 0) {
        if constexpr (NOPS < 0)
            asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
        else if constexpr (NOPS == 0)
            asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\ts_nop 0\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
        else if constexpr (NOPS == 1)
            asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\ts_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
        else if constexpr (NOPS == 2)
            asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\ts_nop 2\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
        else if constexpr (NOPS == 3)
            asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\ts_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
        else if constexpr (NOPS == 5)
            asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\ts_nop 5\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
        else
            asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
    } else if constexpr (FILL == 2) {   // one independent vector instruction in between
        asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\tv_mov_b32 v104, %6\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103", "v104");
    } else if constexpr (FILL == 3) {   // one scalar instruction in between
        asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\ts_mov_b32 s40, 0\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103", "s40");
    } else if constexpr (FILL == 4) {   // an LDS read in between
        asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\tds_read_b32 v104, %9\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(a), "v"(x), "v"(y), "v"(other), "v"(0) : "v100", "v101", "v102", "v103", "v104");
    } else {
        // cvt writes word 0 of b; one independent MFMA (operands a, a) on another accumulator; then the MFMA that reads b
        asm volatile(PRE "v_cvt_pk_bf16_f32 v100, %6, %7\n\tv_mfma_f32_32x32x16_bf16 %8, %5, %5, %8\n\tv_mfma_f32_32x32x16_bf16 %0, %5, v[100:103], %0" OPS : "v100", "v101", "v102", "v103");
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc), "+v"(other));
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = acc[i];
}

int main() {
    float h[640];
    for (int i = 0; i < 640; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
    float *din, *dout;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    float ref[1024], got[1024];
    auto run = [&](auto kern, float* dst) {
        hipMemset(dout, 0, 4096);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, din, dout);
        hipMemcpy(dst, dout, 4096, hipMemcpyDeviceToHost);
    };
    run(k<15, 0>, ref);   // 16 wait states: the reference
    auto report = [&](const char* what) {
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
        printf("%-44s %s (%d of 1024 values differ)\n", what, bad ? "WRONG" : "ok", bad);
    };
    run(k<-1, 0>, got); report("cvt -> mfma, nothing in between");
    run(k<0, 0>, got);  report("cvt -> s_nop 0 -> mfma (1 wait state)");
    run(k<1, 0>, got);  report("cvt -> s_nop 1 -> mfma (2 wait states)");
    run(k<2, 0>, got);  report("cvt -> s_nop 2 -> mfma (3 wait states)");
    run(k<3, 0>, got);  report("cvt -> s_nop 3 -> mfma (4 wait states)");
    run(k<5, 0>, got);  report("cvt -> s_nop 5 -> mfma (6 wait states)");
    run(k<0, 1>, got);  report("cvt -> independent mfma -> mfma");
    run(k<0, 2>, got);  report("cvt -> independent v_mov -> mfma");
    run(k<0, 3>, got);  report("cvt -> s_mov -> mfma");
    run(k<0, 4>, got);  report("cvt -> ds_read -> mfma");
    return 0;
}
