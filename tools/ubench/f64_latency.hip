// Diagnostic micro-benchmark (not part of the product): issue/dependency cost of fp64 VALU chains on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o f64_latency tools/ubench/f64_latency.hip && ./f64_latency
// One wave per workgroup; each test runs NITER steps of K independent dependent-chains and reports
// s_memtime ticks per instruction.  K = 1 is the pure dependent latency, large K the issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NITER 16384

template <int K, int OP>
__global__ void chain(double* out, unsigned long long* ticks, double seed) {
    double x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = seed + k + threadIdx.x * 1e-3;
    const double c1 = 0.999999, c2 = 1e-9;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int i = 0; i < NITER; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c2));
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c1));
            if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[k]));
            if (OP == 4) { float y = (float)x[k]; asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(0.5f)); x[k] = y; }
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int K, int OP>
static void run(const char* name, int blocks, int waves_per_block) {
    double* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(double) * 64 * 16 * blocks);
    hipMalloc(&ticks, sizeof(unsigned long long) * 2 * blocks);
    hipLaunchKernelGGL((chain<K, OP>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, ticks, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<K, OP>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, ticks, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("    [whole launch %.3f ms: %.1f G lane-instructions/s per SIMD-equivalent of 1024 => %.2f T lane-ops/s]\n", ms,
           (double)blocks * waves_per_block * 64.0 * NITER * K / (ms * 1e-3) / 1e9 / 1024.0, (double)blocks * waves_per_block * 64.0 * NITER * K / (ms * 1e-3) / 1e12);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double sum = 0, rsum = 0; for (int i = 0; i < blocks; ++i) { sum += (double)h[2 * i]; rsum += (double)h[2 * i + 1]; }
    printf("%-10s K=%d waves/WG=%d blocks=%d : %.2f ticks per instruction (%.2f per step of K); memtime runs at %.1f MHz (vs 100 MHz realtime); %.3f ns per instruction\n",
           name, K, waves_per_block, blocks, sum / blocks / NITER / K, sum / blocks / NITER, 100.0 * sum / rsum, rsum / blocks * 10.0 / NITER / K);
    hipFree(out); hipFree(ticks);
}

int main() {
    // calibrate ticks: a long kernel timed with events
    {
        double* out; unsigned long long* ticks; hipMalloc(&out, 8 * 64 * 8); hipMalloc(&ticks, 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((chain<8, 0>), dim3(1), dim3(64), 0, 0, out, ticks, 1.0); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL((chain<8, 0>), dim3(1), dim3(64), 0, 0, out, ticks, 1.0); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("calibration: %llu ticks in %.4f ms (event-timed, includes launch) => >= %.3f ticks/ns\n", t, ms, t / (ms * 1e6));
    }
    const char* names[] = {"fma_f64", "add_f64", "mul_f64", "rcp_f64", "cvt+fma32"};
    run<1, 0>(names[0], 1, 1); run<2, 0>(names[0], 1, 1); run<4, 0>(names[0], 1, 1); run<8, 0>(names[0], 1, 1);
    run<1, 1>(names[1], 1, 1); run<2, 1>(names[1], 1, 1); run<4, 1>(names[1], 1, 1); run<8, 1>(names[1], 1, 1);
    run<1, 2>(names[2], 1, 1); run<4, 2>(names[2], 1, 1);
    run<1, 3>(names[3], 1, 1); run<2, 3>(names[3], 1, 1); run<4, 3>(names[3], 1, 1);
    // two / four waves on one SIMD?  4 waves of one WG land on the 4 SIMDs; 8 waves -> 2 per SIMD
    run<1, 0>(names[0], 1, 4); run<1, 0>(names[0], 1, 8);
    run<8, 0>(names[0], 1, 4); run<8, 0>(names[0], 1, 8); run<8, 0>(names[0], 1, 16);
    run<8, 1>(names[1], 1, 4); run<8, 1>(names[1], 1, 8); run<8, 1>(names[1], 1, 16); run<8, 4>(names[4], 1, 4); run<8, 4>(names[4], 1, 16);
    // whole chip busy (DVFS): every CU running 8 waves of the K=8 fma chain
    run<8, 0>(names[0], 1024, 8); run<8, 1>(names[1], 1024, 8); run<8, 0>(names[0], 1024, 4); run<8, 0>(names[0], 2048, 4); run<8, 2>(names[2], 1024, 8);
    return 0;
}
