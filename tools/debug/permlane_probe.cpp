// What v_permlane32_swap_b32 returns when both operands hold the lane id (test infrastructure).
// hipcc --offload-arch=gfx950 -O3 tools/debug/permlane_probe.cpp -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    const unsigned v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(v, v + 100u, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc((void**)&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("first operand = lane, second = lane + 100\nr[0]:"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]);
    printf("\nr[1]:"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]);
    printf("\n");
    return 0;
}
