// Diagnostic micro-benchmark (not part of the product): issue rate of the sliding-window fma block of scr_refine_kernel
// (16 accumulators, a 31-value window and 16 taps, all in registers: 256 v_fmac_f64 with three distinct 64-bit register
// operands each) at 1, 2 and more waves per SIMD, against the same count of fma on a constant pair.
//   hipcc -O3 --offload-arch=gfx950 -o fma_block tools/ubench/fma_block.hip && ./fma_block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REPS 2048

template <int MODE>
__global__ __launch_bounds__(256) void blockk(double* out, unsigned long long* ticks, double seed) {
    double acc[16], x[31], c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = 0.0; c[j] = seed * (j + 1) + threadIdx.x * 1e-6; }
#pragma unroll
    for (int i = 0; i < 31; ++i) x[i] = 1.0 / (seed + i + threadIdx.x);
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (MODE == 0) acc[j] = fma(c[kk], x[kk + j], acc[j]);                 // the refine block
                if (MODE == 1) acc[j] = fma(c[0], x[0], acc[j]);                       // same count, one operand pair
                if (MODE == 2) acc[j] = fma(c[kk], x[j], acc[j]);                      // window does not slide
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(acc[j]));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int blocks) {
    double* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(double) * 256 * blocks);
    hipMalloc(&ticks, sizeof(unsigned long long) * blocks);
    hipLaunchKernelGGL((blockk<MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, 1.5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((blockk<MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, 1.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < blocks; ++i) sum += (double)h[i];
    const double fma_per_wave = 256.0 * REPS;
    printf("%-28s workgroups %5d (%.1f waves per SIMD): %.3f ms; %.2f ticks per fma and wave; %.2f T fma lane-ops/s = %.1f lanes per clock and SIMD at 2.4 GHz\n",
           name, blocks, blocks * 4 / 1024.0, ms, sum / blocks / fma_per_wave, blocks * 256.0 * fma_per_wave / (ms * 1e-3) / 1e12,
           blocks * 256.0 * fma_per_wave / (ms * 1e-3) / 1024.0 / 2.4e9);
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int blocks : {256, 512, 768, 1024}) {
        run<0>("sliding window (refine)", blocks);
        run<1>("one operand pair", blocks);
        run<2>("fixed window", blocks);
    }
    return 0;
}
