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

// MODE 1: + one ds_read_b64 per MFMA (asm, waited 8 behind); 2: + one buffer_load_dword ... lds per MFMA; 3: both
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(float* out, const float* src, unsigned long long* clk, int iters) {
    __shared__ float lds[16384];
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1.0f;
    __syncthreads();
    const unsigned laddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds + (threadIdx.x & 63) * 8u;
    const unsigned long long a64 = (unsigned long long)src;
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a64);
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a64 >> 32) & 0xffffu));
    rs[2] = 1 << 20;
    rs[3] = 0x00020000;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned m0base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds + 32768u + wave * 4096u;
    const unsigned voff = (threadIdx.x & 63) * 4u;
    f32x2 q[4];
    for (int z = 0; z < 4; ++z) q[z] = f32x2{1.f, 1.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE & 1) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[r & 3]) : "v"(laddr), "n"(512 * 0));
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(q[(r + 2) & 3]));
            }
            if (MODE & 2) {
                const unsigned soff = (unsigned)(((i * 8 + r) & 1023) * 256);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(m0base + (unsigned)(r * 256)), "v"(voff), "s"(rs), "s"(soff) : "memory");
            }
            acc[r & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[(r + 2) & 3][0], q[(r + 2) & 3][1], acc[r & 1], 0, 0, 0);
        }
        if (MODE & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int qq = 0; qq < 16; ++qq) s += acc[c][qq];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x + 8192];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run_mix(const char* name) {
    float *out, *src;
    unsigned long long* clk;
    hipMalloc((void**)&out, 256 * 256 * 4);
    hipMalloc((void**)&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    hipMalloc((void**)&clk, 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mix_kernel<MODE>), dim3(256), dim3(256), 0, 0, out, src, clk, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<MODE>), dim3(256), dim3(256), 0, 0, out, src, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    printf("%-28s %6.1f ns/MFMA/wave  (%.1f counter ticks)\n", name, ms * 1e6 / n, (double)h / n);
}

// the spectral match's register shape: 16 long-lived accumulators (256 registers, AGPRs) updated once per 'step' from two
// short-lived ones that take 64 MFMAs per step
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void shape_kernel(float* out, unsigned long long* clk, int iters) {
    f32x16 acc2[NT];
    for (int c = 0; c < NT; ++c)
        for (int q = 0; q < 16; ++q) acc2[c][q] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        f32x16 ca, cb;
        for (int q = 0; q < 16; ++q) { ca[q] = 0.f; cb[q] = 0.f; }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            ca = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, cb, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[r] + cb[r], b, acc2[r % NT], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < NT; ++c)
        for (int q = 0; q < 16; ++q) s += acc2[c][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NT>
void run_shape(const char* name) {
    float* out;
    unsigned long long* clk;
    hipMalloc((void**)&out, 256 * 256 * 4);
    hipMalloc((void**)&clk, 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(shape_kernel<NT>, dim3(256), dim3(256), 0, 0, out, clk, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(shape_kernel<NT>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 80;
    printf("%-28s %6.1f ns/MFMA/wave  (%.1f counter ticks)\n", name, ms * 1e6 / n, (double)h / n);
}

int main() {
    run_shape<16>("shape: 2 + 16 accumulators");
    run_shape<14>("shape: 2 + 14 accumulators");
    run_shape<8>("shape: 2 + 8 accumulators");
    run_mix<0>("mix: MFMA only");
    run_mix<1>("mix: + ds_read_b64 each");
    run_mix<2>("mix: + LDS-DMA each");
    run_mix<3>("mix: + both");
    run<1, 4>("1 chain, 4 waves/CU");
    run<2, 4>("2 chains, 4 waves/CU");
    run<4, 4>("4 chains, 4 waves/CU");
    run<8, 4>("8 chains, 4 waves/CU");
    run<2, 8>("2 chains, 8 waves/CU");
    run<4, 8>("4 chains, 8 waves/CU");
    return 0;
}
