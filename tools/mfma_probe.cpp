// Diagnostic micro-benchmark (not part of the product): cycles per v_mfma_f32_32x32x2_f32 under
// the conditions of the conv main loop, to separate what costs issue slots from what does not.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.cpp -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = f32x4{g[i & 255], g[(i + 1) & 255], g[(i + 2) & 255], g[(i + 3) & 255]};
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 fa[4], fb[2];
    for (int i = 0; i < 4; ++i) fa[i] = lds[tid + 256 * i];
    for (int i = 0; i < 2; ++i) fb[i] = lds[tid + 256 * (4 + i)];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // fixed operands
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][0], fb[0][0], acc[k % NACC], 0, 0, 0);
        } else {           // operand pattern of the conv tap: 4 A x 2 B fragments x 4 k-steps
            f32x4 na[4], nb[2];
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) na[i] = lds[(tid + 256 * i + it * 64) & 4095];
#pragma unroll
                for (int i = 0; i < 2; ++i) nb[i] = lds[(tid + 256 * (4 + i) + it * 64) & 4095];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[(m * 2 + n) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][j], fb[n][j], acc[(m * 2 + n) % NACC], 0, 0, 0);
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 20; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = na[i];
#pragma unroll
                for (int i = 0; i < 2; ++i) fb[i] = nb[i];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (MODE == 3 || MODE == 4 || MODE == 5) {
        // handled below (kept separate so that MODE 0-2 codegen is untouched)
    }
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

// MODE 3: 6 ds_read_b128 per 32 MFMA into registers no MFMA reads (pure LDS issue cost)
// MODE 4: 12 ds_read_b64 instead      MODE 5: 24 v_mov (VALU only)     MODE 6: 6 ds_read_b128, consumed NEXT-next tap (3 sets)
template <int MODE>
__global__ __launch_bounds__(256) void probe2(const float* __restrict__ g, float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = f32x4{g[i & 255], g[(i + 1) & 255], g[(i + 2) & 255], g[(i + 3) & 255]};
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 fa[4], fb[2], sink[6];
    for (int i = 0; i < 4; ++i) fa[i] = lds[tid + 256 * i];
    for (int i = 0; i < 2; ++i) fb[i] = lds[tid + 256 * (4 + i)];
    for (int i = 0; i < 6; ++i) sink[i] = fa[i & 3];
    const f32x4* base = lds + tid;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 6; ++i) sink[i] = base[256 * i + (it & 1) * 64];
        } else if (MODE == 4) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f32x2 lo = *reinterpret_cast<const f32x2*>(base + 256 * i + (it & 1) * 64);
                f32x2 hi = *(reinterpret_cast<const f32x2*>(base + 256 * i + (it & 1) * 64) + 1);
                sink[i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 6; ++i) sink[i] = sink[i] * 1.0001f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][j], fb[n][j], acc[m * 2 + n], 0, 0, 0);
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < (MODE == 3 ? 6 : 12); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" ::"v"(sink[i]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 6; ++i) s += sink[i][0];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

// MODE 6: conv-like: two fragment sets, reads of set B issued during MFMAs on set A and vice versa,
// constant LDS addresses (no address VALU, no copies). MODE 7: same + 13 global_load_dwordx4 per 9 taps.
template <int MODE>
__global__ __launch_bounds__(256) void probe3(const float* __restrict__ g, float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = f32x4{g[i & 255], g[(i + 1) & 255], g[(i + 2) & 255], g[(i + 3) & 255]};
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 fa[2][4], fb[2][2];
    const f32x4* base = lds + tid;
    for (int i = 0; i < 4; ++i) fa[0][i] = base[256 * i];
    for (int i = 0; i < 2; ++i) fb[0][i] = base[256 * (4 + i)];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x4 stg[13];
    for (int i = 0; i < 13; ++i) stg[i] = fa[0][i & 3];
    const f32x4* gsrc = reinterpret_cast<const f32x4*>(g) + (tid & 63);
    f32x4* wdst = lds + 2048 + tid;
    for (int it = 0; it < iters; it += 2) {
        if ((it & 7) == 0) {   // once per 8 taps: the staging work of one K chunk
            if (MODE == 7 || MODE == 10) {
#pragma unroll
                for (int i = 0; i < 13; ++i) wdst[(i & 3) * 256] = stg[i];
            }
            if (MODE == 8 || MODE == 10) {
#pragma unroll
                for (int i = 0; i < 13; ++i) stg[i] = gsrc[i * 64];
            }
            if (MODE == 9 || MODE == 10) __syncthreads();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[h ^ 1][i] = base[256 * i + 64 * (h + 1)];
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[h ^ 1][i] = base[256 * (4 + i) + 64 * (h + 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][m][j], fb[h][n][j], acc[m * 2 + n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 20; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 13; ++i) s += stg[i][1];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
void run3(const char* name, int blocks, const float* g, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe3<MODE>), dim3(blocks), dim3(256), 0, 0, g, out, cyc, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double per = (double)h[h.size() / 2] / (iters * 32.0);
    double tf = (double)blocks * 4 * iters * 32 * 4096.0 / (best * 1e-3) / 1e12;
    printf("%-44s blocks=%4d  ticks/MFMA %.2f   %.1f TF/s  (%.3f ms)\n", name, blocks, per, tf, best);
}

template <int MODE>
void run2(const char* name, int blocks, const float* g, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe2<MODE>), dim3(blocks), dim3(256), 0, 0, g, out, cyc, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double per = (double)h[h.size() / 2] / (iters * 32.0);
    double tf = (double)blocks * 4 * iters * 32 * 4096.0 / (best * 1e-3) / 1e12;
    printf("%-44s blocks=%4d  ticks/MFMA %.2f   %.1f TF/s  (%.3f ms)\n", name, blocks, per, tf, best);
}

template <int MODE, int NACC>
void run(const char* name, int blocks, const float* g, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, g, out, cyc, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double per = (double)h[h.size() / 2] / (iters * 32.0);
    double tf = (double)blocks * 4 * iters * 32 * 4096.0 / (best * 1e-3) / 1e12;
    printf("%-44s blocks=%4d  ticks/MFMA %.2f   %.1f TF/s  (%.3f ms)\n", name, blocks, per, tf, best);
}

int main() {
    float* g; float* out; unsigned long long* cyc;
    hipMalloc(&g, 4096 * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 4 * 8);
    std::vector<float> h(4096);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int blocks : {512}) {
        run<0, 8>("fixed operands, 8 acc", blocks, g, out, cyc);
        run<0, 4>("fixed operands, 4 acc", blocks, g, out, cyc);
        run<1, 8>("tap operand pattern, regs only, 8 acc", blocks, g, out, cyc);
        run<2, 8>("tap pattern + 6 ds_read_b128 per 32 MFMA", blocks, g, out, cyc);
        run2<3>("6 ds_read_b128 into unused regs", blocks, g, out, cyc);
        run2<4>("12 ds_read_b64 into unused regs", blocks, g, out, cyc);
        run2<5>("24 v_mul VALU only", blocks, g, out, cyc);
        run3<6>("2 sets, 6 ds_read_b128 consumed next tap", blocks, g, out, cyc);
        run3<7>("  + 13 ds_write_b128 per 8 taps", blocks, g, out, cyc);
        run3<8>("  + 13 global_load_dwordx4 per 8 taps", blocks, g, out, cyc);
        run3<9>("  + 1 __syncthreads per 8 taps", blocks, g, out, cyc);
        run3<10>("  + all three", blocks, g, out, cyc);
    }
    return 0;
}
