// Diagnostic micro-benchmark (not part of the product): issue rate of PACKED fp32 VALU instructions against their scalar
// forms on gfx950 -- the question behind a packed-math rebuild of the fp32 screening transform (scr_fft4096): is
// v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (two fp32 operations per lane and instruction, with op_sel / neg operand
// modifiers as a complex multiply needs them) issued at the rate of v_fma_f32, or at half of it?
//   hipcc -O3 --offload-arch=gfx950 -o pk_f32_rate tools/ubench/pk_f32_rate.hip ; ./pk_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define NITER 2048
#define K 12
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void spin(float* out, unsigned long long* ticks) {
    v2f x[K];
    const v2f c1 = {0.999999f, 1.000001f}, c2 = {1e-9f, -1e-9f};
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = v2f{1.0f + k + threadIdx.x * 1e-3f, 2.0f + k};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < NITER; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k].x) : "v"(c1.x), "v"(c2.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k].y) : "v"(c1.y), "v"(c2.y)); }
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
            if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(c2));
            if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(c1));
            if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "+v"(x[k]) : "v"(c1), "v"(c2));
            if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(x[k]) : "v"(c2));
            if (OP == 6) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k].x) : "v"(c2.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k].y) : "v"(c2.y)); }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += x[k].x + x[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int OP> static void run(const char* name, int wg_per_cu) {
    const int blocks = 256 * wg_per_cu;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(float) * 256 * blocks); hipMalloc(&ticks, 8 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin<OP>, dim3(blocks), dim3(256), 0, 0, out, ticks);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(spin<OP>, dim3(blocks), dim3(256), 0, 0, out, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    const double pairs = (double)reps * blocks * 256.0 * NITER * K;          // lane-level PAIRS of fp32 operations
    const double instr_per_wave = (double)NITER * K * ((OP == 0 || OP == 6) ? 2 : 1);
    printf("%-44s %d waves/SIMD: %7.2f T lane-pairs/s   %5.2f cycles per wave-instruction (s_memtime, one workgroup's view)\n", name, wg_per_cu,
           pairs / (ms * 1e-3) / 1e12, (double)h / instr_per_wave * 1.0);
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) { run<0>("2 x v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<4>("v_pk_fma_f32 with op_sel", 1); run<6>("2 x v_add_f32", 1); run<2>("v_pk_add_f32", 1); run<5>("v_pk_add_f32 op_sel + neg", 1); run<3>("v_pk_mul_f32", 1); }
        if (w == 2) { run<0>("2 x v_fma_f32", 2); run<1>("v_pk_fma_f32", 2); run<4>("v_pk_fma_f32 with op_sel", 2); run<6>("2 x v_add_f32", 2); run<2>("v_pk_add_f32", 2); run<5>("v_pk_add_f32 op_sel + neg", 2); run<3>("v_pk_mul_f32", 2); }
        if (w == 4) { run<0>("2 x v_fma_f32", 4); run<1>("v_pk_fma_f32", 4); run<2>("v_pk_add_f32", 4); }
        if (w == 8) { run<0>("2 x v_fma_f32", 8); run<1>("v_pk_fma_f32", 8); run<2>("v_pk_add_f32", 8); }
    }
    return 0;
}
