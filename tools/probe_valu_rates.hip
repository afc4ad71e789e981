// micro-benchmark: issue rate of v_exp_f32 / v_exp_f16 / v_rcp_f32 / v_add_f32 / v_pk_add_f32 / v_cvt_pk_f16_f32 on gfx950 (one wave per SIMD x 4)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 0.001f * threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(double*)&a[i & 6]));
            if (OP == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(a[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 4), dim3(256), 0, 0, d, 1000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 4), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 1024 workgroups x 4 waves = 4096 waves = 4 waves per SIMD (1024 SIMDs); instructions per SIMD = 4 * iters * 8
    printf("%-22s %8.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / (4.0 * iters * 8));
}
int main() {
    float* d;
    hipMalloc(&d, 1024 * 256 * 4);
    run<3>("v_add_f32", d);
    run<0>("v_exp_f32", d);
    run<1>("v_exp_f16", d);
    run<6>("v_exp_f16_sdwa(hi)", d);
    run<2>("v_rcp_f32", d);
    run<4>("v_pk_add_f32", d);
    run<5>("v_cvt_pk_f16_f32", d);
    run<7>("v_pk_mul_f16", d);
    run<3>("v_add_f32", d);
    return 0;
}
