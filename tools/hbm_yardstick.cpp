// HBM yardstick for the HBM-bound kernels' fractions: what a plain 16-byte-per-lane streaming kernel
// reaches on THIS box -- copy (read + write), read-only (sum), write-only (fill) -- over grid sizes, block sizes, unroll depths and
// cache policies. /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; torch's copy_ on this pool gave
// 4.7-5.0 (profiles/r05_hbm_kernels.txt), so fractions "of the copy rate" need this number, not torch's.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_yardstick.cpp -o tools/bin/hbm_yardstick && tools/bin/hbm_yardstick
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
    // grid-stride over chunks of U * blockDim.x vectors: U independent 16-byte loads in flight per lane, then U stores
    const size_t chunk = (size_t)blockDim.x * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
        }
    }
}

template <int U, bool NT>
__global__ void read_kernel(const f4* __restrict__ src, float* __restrict__ out, size_t n) {
    const size_t chunk = (size_t)blockDim.x * U;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            v[u] = (i < n) ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) out[0] = s;          // never true on the test data: keeps the loads alive without a store stream
}

template <bool NT>
__global__ void fill_kernel(f4* __restrict__ dst, size_t n, float val) {
    f4 v = {val, val, val, val};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}

template <typename F>
static double timed(F launch, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms * 1e-3 / iters;
}

int main(int argc, char** argv) {
    size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 2048;       // bytes per buffer, MiB (past the 256 MiB Infinity Cache by default)
    size_t bytes = mb << 20, n = bytes / sizeof(f4);
    f4 *a, *b;
    float* out;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 0, bytes));
    printf("buffers: 2 x %zu MiB; rates in TB/s of bytes moved (copy = read + written); peak 8.0 spec, guide: 6.29 measured float4 copy\n", mb);
    double best_copy = 0, best_read = 0, best_fill = 0;
    const int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384};
    const int blocks[] = {256, 512, 1024};
    for (int blk : blocks)
        for (int g : grids) {
#define RUN_COPY(U, NT) { double t = timed([&] { hipLaunchKernelGGL((copy_kernel<U, NT>), dim3(g), dim3(blk), 0, 0, a, b, n); }, 10); \
            double r = 2.0 * bytes / t / 1e12; if (r > best_copy) best_copy = r; \
            printf("copy  block %4d grid %5d unroll %d %s: %.2f TB/s\n", blk, g, U, NT ? "nt     " : "default", r); }
            RUN_COPY(1, false) RUN_COPY(4, false) RUN_COPY(8, false) RUN_COPY(4, true)
#define RUN_READ(U, NT) { double t = timed([&] { hipLaunchKernelGGL((read_kernel<U, NT>), dim3(g), dim3(blk), 0, 0, a, out, n); }, 10); \
            double r = 1.0 * bytes / t / 1e12; if (r > best_read) best_read = r; \
            printf("read  block %4d grid %5d unroll %d %s: %.2f TB/s\n", blk, g, U, NT ? "nt     " : "default", r); }
            RUN_READ(4, false) RUN_READ(8, false) RUN_READ(8, true)
#define RUN_FILL(NT) { double t = timed([&] { hipLaunchKernelGGL((fill_kernel<NT>), dim3(g), dim3(blk), 0, 0, b, n, 1.0f); }, 10); \
            double r = 1.0 * bytes / t / 1e12; if (r > best_fill) best_fill = r; \
            printf("fill  block %4d grid %5d          %s: %.2f TB/s\n", blk, g, NT ? "nt     " : "default", r); }
            RUN_FILL(false) RUN_FILL(true)
        }
    // one-shot grid (one chunk per block, no loop): the shape of torch's elementwise kernels
    {
        int blk = 256;
        size_t g = (n + (size_t)blk * 4 - 1) / ((size_t)blk * 4);
        double t = timed([&] { hipLaunchKernelGGL((copy_kernel<4, false>), dim3((unsigned)g), dim3(blk), 0, 0, a, b, n); }, 10);
        printf("copy  block  256 grid %zu (one chunk per block) unroll 4: %.2f TB/s\n", g, 2.0 * bytes / t / 1e12);
    }
    {
        double t = timed([&] { CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
        printf("hipMemcpyAsync device-to-device: %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    }
    printf("BEST copy %.2f TB/s (%.3f of 8), read %.2f TB/s (%.3f), fill %.2f TB/s (%.3f)\n", best_copy, best_copy / 8, best_read, best_read / 8,
           best_fill, best_fill / 8);
    return 0;
}
