// Diagnostic micro-benchmark (not part of the product): sustained rate, shader clock and -- together with a
// `rocm-smi --showpower` sampler running beside it (tools/ubench/op_energy.sh) -- energy per operation of the
// instruction classes the receive kernels are made of, with every CU busy.
//   hipcc -O3 --offload-arch=gfx950 -o op_energy tools/ubench/op_energy.hip ; ./op_energy <op> <seconds>
// ops: nop fma64 add64 mul64 addu32 cndmask cvt ldsr ldsw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define NITER 4096
#define K 8
typedef double v2d __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void spin(double* out, unsigned long long* ticks, double seed) {
    __shared__ double4 lds[2048];
    double x[K];
    unsigned int u[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = seed + k + threadIdx.x * 1e-3; u[k] = threadIdx.x * 7 + k; }
    const double c1 = 0.999999, c2 = 1e-9;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_double4(seed, 1.0, 2.0, 3.0);
    __syncthreads();
    const int la = (threadIdx.x * 16) & 0x7ff0;          // 16-byte aligned, conflict-free per wave
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int i = 0; i < NITER; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c2));
            if (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c1));
            if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % K]));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) % K]) : "vcc");
            if (OP == 6) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(x[k])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[k]) : "v"(f)); }
            if (OP == 7) { v2d v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(la + 16 * 256 * (k & 3)) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x[k] += v.x; }
            if (OP == 8) { v2d v = {x[k], x[k]}; asm volatile("ds_write_b128 %0, %1" :: "v"(la + 16 * 256 * (k & 3)), "v"(v) : "memory"); }
        }
        if (OP == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (OP == 0) __builtin_amdgcn_s_sleep(8);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += x[k] + u[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP> static void run(const char* name, double seconds) {
    const int blocks = 256 * 8;                                  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    double* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(double) * 256 * blocks); hipMalloc(&ticks, 16 * blocks);
    hipLaunchKernelGGL(spin<OP>, dim3(blocks), dim3(256), 0, 0, out, ticks, 1.0); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now(); long launches = 0; double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin<OP>, dim3(blocks), dim3(256), 0, 0, out, ticks, 1.0);
        hipDeviceSynchronize(); launches += 20;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    unsigned long long h[2]; hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    const double ops = (double)launches * blocks * 256.0 * NITER * K * (OP == 6 ? 2 : 1);
    printf("%-8s %.2f s: %.2f T lane-ops/s, shader clock %.0f MHz\n", name, el, ops / el / 1e12, 100.0 * h[0] / h[1]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const char* op = argc > 1 ? argv[1] : "fma64";
    const double sec = argc > 2 ? atof(argv[2]) : 4.0;
    if (!strcmp(op, "nop")) run<0>(op, sec); else if (!strcmp(op, "fma64")) run<1>(op, sec); else if (!strcmp(op, "add64")) run<2>(op, sec);
    else if (!strcmp(op, "mul64")) run<3>(op, sec); else if (!strcmp(op, "addu32")) run<4>(op, sec); else if (!strcmp(op, "cndmask")) run<5>(op, sec);
    else if (!strcmp(op, "cvt")) run<6>(op, sec); else if (!strcmp(op, "ldsr")) run<7>(op, sec); else if (!strcmp(op, "ldsw")) run<8>(op, sec);
    else { fprintf(stderr, "unknown op %s\n", op); return 2; }
    return 0;
}
