// Standalone MFMA throughput probe for gfx950 ("settle the ceiling").
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Every wave runs a register-resident chain of v_mfma_f32_32x32x16_{bf16,f16} (ACC independent accumulators, no memory
// traffic); the grid fills every SIMD with WAVES waves.  Reported: TFLOP/s from HIP events, and the sustained shader clock
// from s_memtime (shader cycles) against wall_clock64 (100 MHz constant clock) inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// RANDOM: operands are pseudo-random values in [-1, 1) that differ per lane and per chain (realistic bit toggling in the MFMA
// datapath -> realistic power); otherwise every lane multiplies the same constant.
template <int ACC, bool F16, bool RANDOM = false>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* sink, unsigned long long* clk) {
    f32x16 acc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    uint4 bits = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3f803f80u, 0x3f803f80u);
    uint4 rb[ACC];
    if constexpr (RANDOM) {
        unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
        auto nxt = [&]() {  // two 16-bit floats in [-1, 1): sign random, exponent <= 0
            h = h * 1664525u + 1013904223u;
            const unsigned m = h >> 9;
            return F16 ? (0x38003800u | (m & 0x83ff83ffu)) : (0x3f003f00u | (m & 0x807f807fu));
        };
#pragma unroll
        for (int a = 0; a < ACC; ++a) rb[a] = make_uint4(nxt(), nxt(), nxt(), nxt());
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < ACC; ++a) {
            if constexpr (RANDOM) bits = rb[a];
            if constexpr (F16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bits), __builtin_bit_cast(f16x8, bits), acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bits), __builtin_bit_cast(bf16x8, bits), acc[a], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < ACC; ++a) s += acc[a][0] + acc[a][15];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <int ACC, bool F16, bool RANDOM = false>
void run(const char* name, int cus, int waves_per_simd, int iters) {
    float* sink;
    unsigned long long* clk;
    hipMalloc(&sink, 16);
    hipMalloc(&clk, 16);
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one wave per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<ACC, F16, RANDOM>), dim3(blocks), dim3(256), 0, 0, iters / 10, sink, clk);  // warm-up (clocks ramp)
    hipDeviceSynchronize();
    float best = 0.0f, ms_sum = 0.0f;
    const int reps = 5;
    unsigned long long h[2] = {0, 0};
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<ACC, F16, RANDOM>), dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms_sum += ms;
        const double tf = 2.0 * 32 * 32 * 16 * (double)ACC * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;
        if (tf > best) best = (float)tf;
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    }
    const double mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;  // wall_clock64 ticks at 100 MHz
    printf("%-44s waves/SIMD %d  acc %d : best %.0f TFLOP/s, mean %.0f (%.2f ms/launch), shader clock in-kernel %.0f MHz\n", name, waves_per_simd, ACC, best,
           2.0 * 32 * 32 * 16 * (double)ACC * iters * 4.0 * blocks / (ms_sum / reps * 1e-3) / 1e12, ms_sum / reps, mhz);
    hipFree(sink);
    hipFree(clk);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int cus = p.multiProcessorCount;
    run<4, false>("v_mfma_f32_32x32x16_bf16 (4 chains/wave)", cus, 1, 20000);
    run<4, false>("v_mfma_f32_32x32x16_bf16 (4 chains/wave)", cus, 2, 10000);
    run<8, false>("v_mfma_f32_32x32x16_bf16 (8 chains/wave)", cus, 2, 5000);
    run<4, true>("v_mfma_f32_32x32x16_f16  (4 chains/wave)", cus, 2, 10000);
    run<8, false>("v_mfma_f32_32x32x16_bf16 (8 chains/wave), long", cus, 2, 50000);
    run<8, false, true>("v_mfma bf16, RANDOM operands (8 chains), long", cus, 2, 50000);
    run<8, true, true>("v_mfma f16,  RANDOM operands (8 chains), long", cus, 2, 50000);
    run<8, false, true>("v_mfma bf16, RANDOM operands (8 chains), 1 wave/SIMD", cus, 1, 50000);
    return 0;
}
