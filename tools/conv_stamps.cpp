// Diagnostic harness (not part of the product): builds conv3x3.hip with -DWITW_STAMPS and prints, for
// one layer shape, where a workgroup spends its cycles (prologue / K loop / epilogue) plus the wall time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWITW_STAMPS -I witw_amd/csrc \
//         tools/conv_stamps.cpp witw_amd/csrc/api.hip -o gpurun_out/conv_stamps
#include "../witw_amd/csrc/conv3x3.hip"
#include <vector>
#include <algorithm>
#include <cstdlib>

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 16, W = argc > 3 ? atoi(argv[3]) : 64;
    int Cin = argc > 4 ? atoi(argv[4]) : 512, Cout = argc > 5 ? atoi(argv[5]) : 512;
    int pool = argc > 6 ? atoi(argv[6]) : 0;
    size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout;
    float *x, *y, *wpk, *bias;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4);
    long long nw = witw_conv3x3_packed_floats(Cout, Cin);
    hipMalloc(&wpk, nw * 4); hipMalloc(&bias, witw_conv3x3_bias_floats(Cout) * 4);
    std::vector<float> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = (float)rand() / RAND_MAX - 0.5f;
    for (long long i = 0; i < nw; ++i) hw[i] = ((float)rand() / RAND_MAX - 0.5f) * 0.05f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(wpk, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, witw_conv3x3_bias_floats(Cout) * 4);
    int TN = witw_conv3x3_tile_n(Cout);
    const int NWV = 8;      // waves per workgroup upper bound; grid upper bound incl. the XCD padding
    long long nblk = (long long)((Cout + TN - 1) / TN) * (8LL * (((long long)B * ((W + 15) / 16) * ((H + 3) / 4) + 7) / 8)) + 64;
    hipMalloc(&witw_conv_stamps_ptr, nblk * NWV * 8 * 8);
    hipMemset(witw_conv_stamps_ptr, 0, nblk * NWV * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        int rc = witw_conv3x3_fwd(x, wpk, bias, nullptr, y, B, H, W, Cin, Cout, 1, 1, 1, pool, 0, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = 2.0 * Cin * Cout * 9 * H * W * B;
        printf("rc=%d  %.3f ms  %.1f TF/s\n", rc, ms, fl / ms / 1e9);
    }
    std::vector<unsigned long long> st(nblk * NWV * 8);
    hipMemcpy(st.data(), witw_conv_stamps_ptr, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> pro, loop, epi, tot, clk;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (long long b = 0; b < nblk * NWV; ++b) {
        unsigned long long* o = &st[b * 8];
        if (o[7] < 4) continue;
        if (o[6]) clk.push_back((double)(o[3] - o[0]) / (double)o[6] * 0.1);
        pro.push_back((double)(o[1] - o[0])); loop.push_back((double)(o[2] - o[1])); epi.push_back((double)(o[3] - o[2]));
        tot.push_back((double)(o[3] - o[0]));
        tmin = std::min(tmin, o[0]); tmax = std::max(tmax, o[3]);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    printf("waves=%zu  median cycles(memtime ticks): prologue %.0f  kloop %.0f  epilogue %.0f  total %.0f   span %.0f\n",
           pro.size(), med(pro), med(loop), med(epi), med(tot), (double)(tmax - tmin));
    printf("in-kernel shader clock (s_memtime / s_memrealtime): median %.3f GHz -> fp32 MFMA peak at that clock %.1f TF/s\n", med(clk),
           157.3 * med(clk) / 2.4);
    int nkc = Cin / 8;
    printf("kloop per chunk %.0f ticks; ideal MFMA per chunk %d cycles\n", med(loop) / nkc, (TN == 128 ? 288 : 144) * 64);
    return 0;
}
