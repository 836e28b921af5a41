// Diagnostic micro-benchmarks behind DESIGN.md's energy budget of demod_kernel (not part of the product).
//
// tools/ubench/op_energy.hip prices an instruction class as (package power - power of a sleeping grid) / rate at FULL
// issue density.  That folds whatever a CU burns by merely being awake (clocks running, waves resident, nothing
// issuing) into the per-operation figure, and a kernel that issues 60 % of the time is then under-priced: the awake
// share is paid for 100 % of the time.  This tool separates the two: the same instruction stream at several issue
// densities (s_nop padding, one or two waves per SIMD, all CUs or a subset) gives package power as a line
//     P(d) = P_awake + d * rate_max * e_op
// whose intercept is the awake power and whose slope is the energy per operation.  It also prices the memory side,
// which op_energy does not have: streaming reads from HBM (a 16 GiB buffer, far past the Infinity Cache), from the
// Infinity Cache (a 128 MiB buffer) and from L2 (1 MiB per XCD), in pJ per byte above the awake power.
//
// Built as a shared library and driven by tools/ubench/energy_budget.py (which samples the amdgpu hwmon power node):
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libenergy.so tools/ubench/energy_budget.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

#define K 8
typedef double v2d __attribute__((ext_vector_type(2)));
// OP: 0 fma64, 1 add64, 2 addu32, 3 ds_read_b128, 4 ds_write_b128, 5 sleep, 6 nothing but the padding,
//     7 mul64, 8 v_cndmask_b32, 9 v_mov_b32, 10 s_add_u32 (scalar unit), 11 v_cvt_f64_f32,
//     12 / 13 / 14: fma64 / add64 / mul64 on operands with RANDOM mantissas, different in every lane and every register
//     15 / 16: ds_read_b128 / ds_write_b128 of random bits (3 and 4 move the same few values);
//     (each chain applies an operation and then its inverse, so the values stay where they are and keep their bits busy;
//     ops 0, 1 and 7 iterate towards one value shared by all lanes and registers, which toggles next to nothing --
//     and power depends on the data, MI355X_MICROARCH.md "DVFS give-back" item 1).  Two instructions per chain and step.
// pad: s_nop 15 (16 idle cycles) repeated `pad` times after every K instructions (K * 4 issue cycles)
template <int OP>
__global__ __launch_bounds__(256) void spin(double* out, unsigned long long* ticks, int iters, int pad, double seed) {
    extern __shared__ double4 lds[];
    double x[K];
    unsigned u[K], sc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = seed + k + threadIdx.x * 1e-3; u[k] = threadIdx.x * 7 + k; sc[k] = blockIdx.x + k; }
    const double c1 = 0.999999, c2 = 1e-9;
    double ra[K], rb[K], rc[K], rd[K];                      // random-mantissa operands: x -> x a + b -> (x a + b) / a - b / a = x
    if (OP >= 12) {
        unsigned long long h = 0x9e3779b97f4a7c15ull * (blockIdx.x * 256ull + threadIdx.x + 1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
            x[k] = 1.0 + (double)(h & 0xfffffffffffffull) * 0x1p-52;
            h ^= h >> 29; h *= 0x94d049bb133111ebull; h ^= h >> 32;
            ra[k] = 1.0 + (double)(h & 0xfffffffffffffull) * 0x1p-52;
            h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
            rb[k] = 0.5 + (double)(h & 0xfffffffffffffull) * 0x1p-53;
            rc[k] = 1.0 / ra[k]; rd[k] = -rb[k] / ra[k];
        }
    }
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_double4(seed, 1.0, 2.0, 3.0);
    if (OP == 15) {                                         // random bits in the LDS image that is read
        unsigned long long h = 0x9e3779b97f4a7c15ull * (blockIdx.x * 256ull + threadIdx.x + 7);
        for (int i = threadIdx.x; i < 1024; i += 256) {
            double v4[4];
            for (int q = 0; q < 4; ++q) { h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32; v4[q] = 1.0 + (double)(h & 0xfffffffffffffull) * 0x1p-52; }
            lds[i] = make_double4(v4[0], v4[1], v4[2], v4[3]);
        }
    }
    __syncthreads();
    const int la = (threadIdx.x * 16) & 0x3ff0;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c2));
            if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % K]));
            if (OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c1));
            if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) % K]) : "vcc");
            if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) % K]));
            if (OP == 10) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc[k]) :: "scc");
            if (OP == 11) { float f = (float)u[k]; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[k]) : "v"(f)); }
            if (OP == 12) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(ra[k]), "v"(rb[k])); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(rc[k]), "v"(rd[k])); }
            if (OP == 13) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(rb[k])); asm volatile("v_add_f64 %0, %0, -%1" : "+v"(x[k]) : "v"(rb[k])); }
            if (OP == 14) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(ra[k])); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(rc[k])); }
            if (OP == 3 || OP == 15) { v2d v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(la + 16 * 256 * (k & 3)) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x[k] += v.x; }
            if (OP == 4) { v2d v = {x[k], x[k]}; asm volatile("ds_write_b128 %0, %1" :: "v"(la + 16 * 256 * (k & 3)), "v"(v) : "memory"); }
            if (OP == 16) { v2d v = {ra[k], rb[(k + i) & (K - 1)]}; asm volatile("ds_write_b128 %0, %1" :: "v"(la + 16 * 256 * (k & 3)), "v"(v) : "memory"); }
        }
        if (OP == 4 || OP == 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (OP == 5) __builtin_amdgcn_s_sleep(127);
        for (int p = 0; p < pad; ++p) asm volatile("s_nop 15");
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += x[k] + u[k] + sc[k];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

// streaming read: every workgroup sums `per_wg` bytes starting at its own offset (16 bytes per lane per load, eight loads
// in flight per thread); `wrap` confines the addresses to a window (L2- or cache-resident variants); XCD-aware when
// xcd_regions: workgroups of one XCD (blockIdx % 8) share one window
__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ in, size_t n16, size_t per_wg16, size_t wrap16, int xcd_regions,
                                                   float* out) {
    const size_t wg = blockIdx.x;
    size_t base = xcd_regions ? (wg & 7) * wrap16 : (wg * per_wg16) % (n16 - per_wg16 + 1);
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = threadIdx.x; i < per_wg16; i += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            size_t o = i + (size_t)j * 256;
            if (xcd_regions) o %= wrap16;
            v[j] = (o < per_wg16 || xcd_regions) ? in[base + o] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[wg] = acc.x;           // (keeps the loads)
}

__global__ void fill_random(unsigned long long* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = 0x9e3779b97f4a7c15ull * (i + 1);
        h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32; h *= 0x94d049bb133111ebull; h ^= h >> 29;
        p[i] = h;
    }
}

// the same with 8-byte loads (what the receive kernels issue: one packed complex point = two f32 samples per lane and load)
__global__ __launch_bounds__(256) void stream_read8(const float2* __restrict__ in, size_t n8, size_t per_wg8, float* out) {
    const size_t wg = blockIdx.x;
    const size_t base = (wg * per_wg8) % (n8 - per_wg8 + 1);
    float2 acc = make_float2(0, 0);
    for (size_t i = threadIdx.x; i < per_wg8; i += 256 * 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const size_t o = i + (size_t)j * 256; v[j] = o < per_wg8 ? in[base + o] : make_float2(0, 0); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; }
    }
    if (acc.x + acc.y == 12345.678f) out[wg] = acc.x;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Result { double seconds, rate, clock_mhz; };

template <int OP> static Result run_spin(int blocks, size_t lds, int iters, int pad, double seconds, double* out, unsigned long long* ticks) {
    hipFuncSetAttribute((const void*)spin<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(spin<OP>, dim3(blocks), dim3(256), lds, 0, out, ticks, iters, pad, 1.0);
    hipDeviceSynchronize();
    const double t0 = now();
    long launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(spin<OP>, dim3(blocks), dim3(256), lds, 0, out, ticks, iters, pad, 1.0);
        hipDeviceSynchronize();
        launches += 8;
        el = now() - t0;
    }
    unsigned long long h[2];
    hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    Result r;
    r.seconds = el;
    r.rate = (double)launches * blocks * 256.0 * iters * K * ((OP >= 12 && OP <= 14) ? 2 : 1) / el;      // lane-operations per second
    r.clock_mhz = 100.0 * (double)h[0] / (double)h[1];
    return r;
}

// op: see spin<>.  blocks: workgroups of 4 waves; lds_bytes: dynamic LDS per workgroup (>= 16384; 81920 leaves one workgroup per
// CU = one wave per SIMD, 40960 two).  Returns 0 and fills out3 = {seconds, lane-ops/s, shader clock MHz}.
extern "C" int eb_spin(int op, int blocks, int lds_bytes, int iters, int pad, double seconds, double* out3) {
    double* out; unsigned long long* ticks;
    if (hipMalloc(&out, sizeof(double) * 256 * (size_t)blocks) != hipSuccess || hipMalloc(&ticks, 16 * (size_t)blocks) != hipSuccess) return -1;
    Result r{};
    switch (op) {
        case 0: r = run_spin<0>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 1: r = run_spin<1>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 2: r = run_spin<2>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 3: r = run_spin<3>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 4: r = run_spin<4>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 5: r = run_spin<5>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 7: r = run_spin<7>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 8: r = run_spin<8>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 9: r = run_spin<9>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 10: r = run_spin<10>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 11: r = run_spin<11>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 12: r = run_spin<12>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 13: r = run_spin<13>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 14: r = run_spin<14>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 15: r = run_spin<15>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        case 16: r = run_spin<16>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
        default: r = run_spin<6>(blocks, lds_bytes, iters, pad, seconds, out, ticks); break;
    }
    hipFree(out); hipFree(ticks);
    out3[0] = r.seconds; out3[1] = r.rate; out3[2] = r.clock_mhz;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Streaming read of `bytes` per launch from a buffer of `buf_bytes` (wrap_bytes > 0: every XCD's workgroups cycle over their own
// window of that size).  out2 = {seconds, bytes/s}.
extern "C" int eb_stream(size_t buf_bytes, size_t bytes_per_launch, size_t wrap_bytes, int blocks, int random_data, double seconds, double* out2) {
    float4* buf; float* out;
    if (hipMalloc(&buf, buf_bytes) != hipSuccess || hipMalloc(&out, 4 * (size_t)blocks) != hipSuccess) return -1;
    if (random_data & 1) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned long long*)buf, buf_bytes / 8);
    else hipMemset(buf, 0, buf_bytes);
    const size_t n16 = buf_bytes / 16, per_wg16 = bytes_per_launch / 16 / blocks, wrap16 = wrap_bytes / 16;
    const bool narrow = random_data >= 2;                   // random_data: 0 zeros, 1 random bits; +2: 8-byte loads instead of 16-byte ones
    auto launch = [&]() {
        if (narrow) hipLaunchKernelGGL(stream_read8, dim3(blocks), dim3(256), 0, 0, (const float2*)buf, n16 * 2, per_wg16 * 2, out);
        else hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, buf, n16, per_wg16, wrap16, wrap_bytes > 0 ? 1 : 0, out);
    };
    launch();
    hipDeviceSynchronize();
    const double t0 = now();
    long launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 4; ++i) launch();
        hipDeviceSynchronize();
        launches += 4;
        el = now() - t0;
    }
    hipFree(buf); hipFree(out);
    out2[0] = el; out2[1] = (double)launches * (double)per_wg16 * 16.0 * blocks / el;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
