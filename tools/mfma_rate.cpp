// Micro-benchmark (test infrastructure): cycles per v_mfma_f32_32x32x2_f32 issued by ONE wave per SIMD, for 1 / 2 / 4 / 8
// independent accumulation chains, with the accumulators left to the compiler (AGPRs under register pressure) --
// the question the match_dft kernel raised. hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.cpp -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rate_kernel(float* out, unsigned long long* clk, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int CH, int WAVES>
void run(const char* name) {
    float* out;
    unsigned long long* clk;
    hipMalloc((void**)&out, 1024 * 64 * WAVES * 4);
    hipMalloc((void**)&clk, 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<CH, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, clk, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<CH, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    printf("%-28s %6.1f ns/MFMA/wave  (%.1f counter ticks)  -> %.1f TF/s on 256 CUs\n", name, ms * 1e6 / n, (double)h / n,
           256.0 * WAVES * n * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(clk);
}

int main() {
    run<1, 4>("1 chain, 4 waves/CU");
    run<2, 4>("2 chains, 4 waves/CU");
    run<4, 4>("4 chains, 4 waves/CU");
    run<8, 4>("8 chains, 4 waves/CU");
    run<2, 8>("2 chains, 8 waves/CU");
    run<4, 8>("4 chains, 8 waves/CU");
    return 0;
}
