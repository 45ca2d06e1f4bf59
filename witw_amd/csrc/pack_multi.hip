// All bf16 filter images of one encoder in ONE launch. A bf16 training step re-packs every trainable filter after each Adam update
// (fp32 master weights -> the bf16 MFMA kernels' filter images: 6 forward + 5 dgrad images per encoder, plus 6 bias copies); done
// one layer at a time that was 24 pack launches of ~11 us and ~35 small copy / fill launches per step, latency-bound: 0.5 ms of the
// 13.2 ms step (profiles/r05_bf16_train_kernel_stats.csv). The image format is conv3x3_bf16.hip's (pack_weights_bf16_kernel):
//   wpk[nt][kc][tap][g][n][0..7] (bf16) <- w[cout][cin][kh][kw] (fp32, torch KCRS), TN = 128 from 128 output channels on, else 64;
//   transpose_flip builds the dgrad filter: (Cout, Cin) describe the PACKED filter, the source is [Cin][Cout][3][3] and
//   w'[co][ci][kh][kw] = w[ci][co][2-kh][2-kw].
// Reference semantics: the weights torch.optim.Adam has just updated (model/cvig_fov.py:416-418, 460) are what the next forward uses.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PM_MAX = 16;

struct PackMultiEntry {
    const float* w;
    unsigned short* wpk;
    const float* bias;        // nullptr: no bias copy
    float* bias_dst;          // [bias_n] floats: the first cout <- bias, the padding stays as it is (zeros from the allocation)
    int cout, cin, n_tiles, nkc, TN, transpose, bias_n;
    unsigned slot_begin;      // first 16-byte slot of this entry in the launch's index space
    unsigned bias_begin;      // first bias element of this entry behind all slots
};

struct PackMultiArgs {
    PackMultiEntry e[PM_MAX];
    int n;
    unsigned total_slots, total;
};

__global__ void pack_weights_bf16_multi_kernel(PackMultiArgs a) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.total) return;
    if (idx >= a.total_slots) {                 // bias copies
        const unsigned b = idx - a.total_slots;
        int k = 0;
#pragma unroll 1
        for (int i = 1; i < a.n; ++i)
            if (b >= a.e[i].bias_begin) k = i;
        const unsigned j = b - a.e[k].bias_begin;
        if (a.e[k].bias != nullptr && j < (unsigned)a.e[k].cout) a.e[k].bias_dst[j] = a.e[k].bias[j];
        return;
    }
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < a.n; ++i)
        if (idx >= a.e[i].slot_begin) k = i;
    const PackMultiEntry& e = a.e[k];
    unsigned t = idx - e.slot_begin;
    const int n = t % e.TN; t /= e.TN;
    const int g = t % 2; t /= 2;
    const int tap = t % 9; t /= 9;
    const int kc = t % e.nkc; t /= e.nkc;
    const int nt = (int)t;
    const int kh = tap / 3, kw = tap % 3;
    const int co = nt * e.TN + n;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = kc * 16 + g * 8 + j;
        float f = 0.f;
        if (co < e.cout && ci < e.cin)
            f = e.transpose ? e.w[(((size_t)ci * e.cout + co) * 3 + (2 - kh)) * 3 + (2 - kw)]
                            : e.w[(((size_t)co * e.cin + ci) * 3 + kh) * 3 + kw];
        v[j] = (__bf16)f;
    }
    reinterpret_cast<bf16x8*>(e.wpk)[idx - e.slot_begin] = v;
}

}  // namespace

extern "C" {

// n filter images in as few launches as their number allows (16 per launch). Entry i: w[i] fp32 [cout][cin][3][3] (transpose[i]:
// [cin][cout][3][3], see above) -> wpk[i] (witw_conv3x3_bf16_packed_elems(cout[i], cin[i]) bf16 elements); bias[i] NULL or fp32
// [cout[i]] -> the first cout[i] floats of bias_dst[i]. The same bits as witw_conv3x3_bf16_pack_weights_ex entry by entry.
int witw_conv3x3_bf16_pack_weights_multi(const void* const* w, void* const* wpk, const void* const* bias, void* const* bias_dst,
                                         const int* cout, const int* cin, const int* transpose, int n, void* stream) {
    WITW_CHECK_ARG(w && wpk && bias && bias_dst && cout && cin && transpose && n >= 0, "bf16 pack_weights_multi: null table / bad count");
    for (int base = 0; base < n; base += PM_MAX) {
        PackMultiArgs a;
        a.n = n - base < PM_MAX ? n - base : PM_MAX;
        unsigned slots = 0, nb = 0;
        for (int i = 0; i < a.n; ++i) {
            const int k = base + i;
            WITW_CHECK_ARG(w[k] && wpk[k] && cout[k] > 0 && cin[k] > 0 && (bias[k] == nullptr || bias_dst[k] != nullptr),
                           "bf16 pack_weights_multi: entry %d is malformed", k);
            PackMultiEntry& e = a.e[i];
            e.w = (const float*)w[k]; e.wpk = (unsigned short*)wpk[k]; e.bias = (const float*)bias[k]; e.bias_dst = (float*)bias_dst[k];
            e.cout = cout[k]; e.cin = cin[k]; e.transpose = transpose[k];
            e.TN = cout[k] >= 128 ? 128 : 64;
            e.n_tiles = cdiv(cout[k], e.TN); e.nkc = cdiv(cin[k], 16);
            e.bias_n = cout[k];
            e.slot_begin = slots; e.bias_begin = nb;
            const unsigned long long s = (unsigned long long)e.n_tiles * e.nkc * 9 * 2 * e.TN;
            WITW_CHECK_ARG(slots + s < 0x7fffffffull, "bf16 pack_weights_multi: too many slots in one launch");
            slots += (unsigned)s;
            nb += (unsigned)cout[k];
        }
        a.total_slots = slots;
        a.total = slots + nb;
        if (a.total == 0) continue;
        hipLaunchKernelGGL(pack_weights_bf16_multi_kernel, dim3((a.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
        WITW_CHECK_LAUNCH("bf16 pack_weights_multi");
    }
    return WITW_OK;
}

}  // extern "C"
