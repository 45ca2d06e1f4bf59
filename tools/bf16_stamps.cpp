// Diagnostic harness (not part of the product): builds conv3x3_bf16.hip with -DWITW_BF_STAMPS and prints, for one layer
// shape, how many s_memtime ticks a wave spends in the K loop, in the vmcnt drain and at the workgroup barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWITW_BF_STAMPS -I witw_amd/csrc \
//         tools/bf16_stamps.cpp witw_amd/csrc/api.hip -o tools/bin/bf16_stamps
#include "../witw_amd/csrc/conv3x3_bf16.hip"
#include <vector>
#include <algorithm>
#include <cstdlib>

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 16, W = argc > 3 ? atoi(argv[3]) : 64;
    int Cin = argc > 4 ? atoi(argv[4]) : 512, Cout = argc > 5 ? atoi(argv[5]) : 512;
    size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout;
    unsigned short *x, *y, *wpk;
    float* bias;
    hipMalloc(&x, nx * 2); hipMalloc(&y, ny * 2);
    long long nw = witw_conv3x3_bf16_packed_elems(Cout, Cin);
    hipMalloc(&wpk, nw * 2); hipMalloc(&bias, 4096 * 4);
    std::vector<unsigned short> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // random bf16 near +-0.01..
    for (long long i = 0; i < nw; ++i) hw[i] = (unsigned short)(0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(wpk, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 4096 * 4);
    const int NWV = 8;
    long long nblk = 8LL * (((long long)B * ((W + 63) / 64) * ((H + 7) / 8) + 7) / 8) * ((Cout + 127) / 128) + 64;
    hipMalloc(&witw_bf16_stamps_ptr, nblk * NWV * 8 * 8);
    hipMemset(witw_bf16_stamps_ptr, 0, nblk * NWV * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        int rc = witw_conv3x3_bf16_fwd(x, wpk, bias, y, B, H, W, Cin, Cout, 1, 1, 1, 0, 0, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rc=%d  %.3f ms  %.1f TF/s\n", rc, ms, 2.0 * Cin * Cout * 9 * H * W * B / ms / 1e9);
    }
    std::vector<unsigned long long> st(nblk * NWV * 8);
    hipMemcpy(st.data(), witw_bf16_stamps_ptr, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> loop, vm, bar, pro, epi, clk;
    for (long long i = 0; i < nblk * NWV; ++i) {
        if (st[i * 8 + 3] != 1) continue;
        loop.push_back((double)st[i * 8]); vm.push_back((double)st[i * 8 + 1]); bar.push_back((double)st[i * 8 + 2]);
        pro.push_back((double)st[i * 8 + 4]); epi.push_back((double)st[i * 8 + 5]);
        if (st[i * 8 + 7]) clk.push_back((double)st[i * 8 + 6] / (double)st[i * 8 + 7] * 0.1);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    auto mean = [](const std::vector<double>& v) { double s = 0; for (double d : v) s += d; return v.empty() ? 0.0 : s / v.size(); };
    int nkc = Cin / 16;
    printf("waves=%zu  per chunk (ticks, 100 MHz?): loop med %.1f mean %.1f | vmcnt wait med %.1f mean %.1f | barrier med %.1f mean %.1f\n",
           loop.size(), med(loop) / nkc, mean(loop) / nkc, med(vm) / nkc, mean(vm) / nkc, med(bar) / nkc, mean(bar) / nkc);
    printf("per workgroup: prologue med %.0f mean %.0f | K loop med %.0f | epilogue med %.0f mean %.0f ticks\n", med(pro), mean(pro), med(loop),
           med(epi), mean(epi));
    printf("in-kernel shader clock (s_memtime / s_memrealtime): median %.3f GHz -> dense bf16 MFMA peak at that clock %.0f TF/s\n", med(clk),
           2500.0 * med(clk) / 2.4);
    printf("ideal MFMA issue per chunk per wave: %d cycles (x2 waves per SIMD)\n", 72 * 32);
    return 0;
}
