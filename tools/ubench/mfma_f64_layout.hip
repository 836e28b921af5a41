// Diagnostic: where v_mfma_f64_16x16x4 keeps its operands and results (gfx950).  A[i][k] = 100 i + k, B[k][j] = delta(k, k0) * (j + 1)
// for k0 = 0..3 -> D[i][j] = (100 i + k0)(j + 1): prints, for every (lane, register), the (i, j) it holds.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f64_layout tools/ubench/mfma_f64_layout.hip && ./mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int assumeA /* lane -> (i = l % 16, k = l / 16) */) {
    const int l = threadIdx.x;
    const int i = l % 16, k = l / 16;
    const double a = 100.0 * i + k;                 // A[i][k] under the assumed operand layout
    for (int k0 = 0; k0 < 4; ++k0) {
        const double b = (k == k0) ? (double)(l % 16 + 1) : 0.0;     // B[k][j] under the assumed layout j = l % 16
        d4 acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(k0 * 64 + l) * 4 + r] = acc[r];
    }
}
int main() {
    double* d; hipMalloc(&d, 4 * 64 * 4 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1);
    double h[4 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_a = 1, ok_b = 1, ok_ops = 1;
    for (int k0 = 0; k0 < 4; ++k0)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const double v = h[(k0 * 64 + l) * 4 + r];
                // decode: v = (100 i + k0)(j + 1) with j = l % 16 if operands are as assumed
                const int j = l % 16;
                const double q = v / (j + 1);
                const int i = (int)((q - k0) / 100.0 + 0.5);
                if (q != 100.0 * i + k0) ok_ops = 0;
                if (i != (l / 16) + 4 * r) ok_a = 0;
                if (i != 4 * (l / 16) + r) ok_b = 0;
                if (k0 == 1 && (l < 2 || l == 17 || l == 63)) printf("lane %2d reg %d: D[%d][%d]\n", l, r, i, j);
            }
    printf("operand layout A[l%%16][l/16], B[l/16][l%%16]: %s\n", ok_ops ? "consistent" : "NOT as assumed");
    printf("D row = (lane>>4) + 4 r : %s\nD row = 4 (lane>>4) + r : %s\n", ok_a ? "yes" : "no", ok_b ? "yes" : "no");
    return 0;
}
