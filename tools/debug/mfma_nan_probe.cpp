// Is v_mfma_f32_32x32x2_f32 row-separable under NaN? A's row 3 (both k) is NaN, everything else 1: which rows of D are NaN?
// hipcc --offload-arch=gfx950 -O3 tools/debug/mfma_nan_probe.cpp -o /tmp/mfma_nan_probe && /tmp/mfma_nan_probe   (test infrastructure)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* o) {
    const int lane = threadIdx.x, l31 = lane & 31, hk = lane >> 5;
    float a = (l31 == 3) ? NAN : 1.f, b = 1.f;
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 16; ++q) o[((q & 3) + 8 * (q >> 2) + 4 * hk) * 32 + l31] = acc[q];
}
int main() {
    float* d; static float h[1024];
    hipMalloc((void**)&d, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    printf("rows of D with a NaN:");
    for (int r = 0; r < 32; ++r) { int n = 0; for (int c = 0; c < 32; ++c) n += isnan(h[r * 32 + c]); if (n) printf(" %d(%d)", r, n); }
    printf("\n");
    return 0;
}
