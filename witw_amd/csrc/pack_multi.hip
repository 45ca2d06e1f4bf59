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
    const float* w;           // SOURCE tensor [A][Bc][3][3] (torch layout): A = cout, Bc = cin of the nn.Conv2d
    unsigned short* wpk;
    const float* bias;        // nullptr: no bias copy
    float* bias_dst;          // the first cout floats <- bias, the padding stays as it is (zeros from the allocation)
    int A, Bc;                // source dimensions
    int cout, cin;            // PACKED filter's dimensions: (A, Bc), or (Bc, A) for the dgrad image
    int nkc, TN, transpose;
    int nb_a, nb_b;           // source blocks of 16 (A) x 64 (Bc)
    unsigned block_begin;     // first workgroup of this entry
};

struct PackMultiArgs {
    PackMultiEntry e[PM_MAX];
    int n;
    unsigned total_blocks;    // workgroups that pack; one more copies the biases
};

// One workgroup per source block of 16 x 64 (A x Bc) filters x 9 taps: the block is 16 runs of 2,304 contiguous bytes (coalesced
// loads into the LDS), from which the threads assemble the 1,152 16-byte slots of that block -- forward image: 16 n x 4 chunks x 9
// taps x 2 groups, written in runs of 256 B; dgrad image (w'[co][ci][kh][kw] = w[ci][co][2-kh][2-kw]): 64 n x 1 chunk x 9 taps x 2
// groups, written in runs of 1 KB. The one-slot-per-thread form gathered 8 floats at a stride of 36 B / Cin * 36 B per thread:
// 103 us per launch for the 11 images of an encoder, against the ~20 us their 87 MB take at HBM speed.
__global__ __launch_bounds__(256) void pack_weights_bf16_multi_kernel(PackMultiArgs a) {
    __shared__ float blk[16][64 * 9 + 1];          // [a][b * 9 + tap], odd row length: the transposed reads stay conflict-free
    const int tid = threadIdx.x;
    if (blockIdx.x >= a.total_blocks) {            // the last workgroup: bias copies
        for (int i = 0; i < a.n; ++i)
            if (a.e[i].bias != nullptr)
                for (int j = tid; j < a.e[i].A; j += 256) a.e[i].bias_dst[j] = a.e[i].bias[j];
        return;
    }
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < a.n; ++i)
        if (blockIdx.x >= a.e[i].block_begin) k = i;
    const PackMultiEntry& e = a.e[k];
    const int lb = blockIdx.x - e.block_begin;
    const int ba = lb % e.nb_a, bb = lb / e.nb_a;
    const int a0 = ba * 16, b0 = bb * 64;
    const int nb = min(64, e.Bc - b0);             // source columns that exist
    // ---- source block -> LDS: row ai = 9 * nb contiguous floats
    for (int i = tid; i < 16 * 576; i += 256) {
        const int ai = i / 576, r = i - ai * 576;
        float v = 0.f;
        if (a0 + ai < e.A && r < nb * 9) v = e.w[((size_t)(a0 + ai) * e.Bc + b0) * 9 + r];
        blk[ai][r] = v;
    }
    __syncthreads();
    bf16x8* out = reinterpret_cast<bf16x8*>(e.wpk);
    if (!e.transpose) {
        // forward image: n = a (16 of them), chunk kc = (b0 + 16 q) / 16 for q = 0..3, group g: 8 input channels b0 + 16 q + 8 g ..
        for (int sl = tid; sl < 16 * 4 * 9 * 2; sl += 256) {
            const int n_l = sl & 15;
            int t = sl >> 4;
            const int g = t & 1; t >>= 1;
            const int tap = t % 9, q = t / 9;
            const int co = a0 + n_l, kc = (b0 >> 4) + q;
            if (kc >= e.nkc || co >= ((e.cout + e.TN - 1) / e.TN) * e.TN) continue;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)blk[n_l][(16 * q + 8 * g + j) * 9 + tap];
            const int nt = co / e.TN, n = co - nt * e.TN;
            out[((((size_t)nt * e.nkc + kc) * 9 + tap) * 2 + g) * e.TN + n] = v;
        }
    } else {
        // dgrad image: packed co = source b (64 of them), packed ci = source a: this block is chunk kc = a0 / 16, groups of 8 a
        for (int sl = tid; sl < 64 * 9 * 2; sl += 256) {
            const int n_l = sl & 63;
            int t = sl >> 6;
            const int g = t & 1;
            const int tap = t >> 1;
            const int co = b0 + n_l, kc = a0 >> 4;
            if (co >= ((e.cout + e.TN - 1) / e.TN) * e.TN) continue;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)blk[8 * g + j][n_l * 9 + (8 - tap)];
            const int nt = co / e.TN, n = co - nt * e.TN;
            out[((((size_t)nt * e.nkc + kc) * 9 + tap) * 2 + g) * e.TN + n] = v;
        }
    }
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
        unsigned blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const int k = base + i;
            WITW_CHECK_ARG(w[k] && wpk[k] && cout[k] > 0 && cin[k] > 0 && (bias[k] == nullptr || bias_dst[k] != nullptr),
                           "bf16 pack_weights_multi: entry %d is malformed", k);
            PackMultiEntry& e = a.e[i];
            e.w = (const float*)w[k]; e.wpk = (unsigned short*)wpk[k]; e.bias = (const float*)bias[k]; e.bias_dst = (float*)bias_dst[k];
            e.cout = cout[k]; e.cin = cin[k]; e.transpose = transpose[k] != 0;
            e.A = e.transpose ? cin[k] : cout[k];            // the source tensor is [A][Bc][3][3]
            e.Bc = e.transpose ? cout[k] : cin[k];
            e.TN = cout[k] >= 128 ? 128 : 64;
            e.nkc = cdiv(cin[k], 16);
            // the image holds cdiv(cout, TN) * TN packed output channels and nkc * 16 input channels, zero beyond the real ones:
            // the blocks cover that padded extent (rows / columns past the source read as zeros)
            const int a_ext = e.transpose ? e.nkc * 16 : cdiv(cout[k], e.TN) * e.TN;
            const int b_ext = e.transpose ? cdiv(cout[k], e.TN) * e.TN : e.nkc * 16;
            e.nb_a = cdiv(a_ext, 16); e.nb_b = cdiv(b_ext, 64);
            e.block_begin = blocks;
            const unsigned long long nbk = (unsigned long long)e.nb_a * e.nb_b;
            WITW_CHECK_ARG(blocks + nbk < 0x7fffffffull, "bf16 pack_weights_multi: too many blocks in one launch");
            blocks += (unsigned)nbk;
            WITW_CHECK_ARG(e.bias == nullptr || !e.transpose, "bf16 pack_weights_multi: entry %d: a dgrad image carries no bias", k);
        }
        a.total_blocks = blocks;
        hipLaunchKernelGGL(pack_weights_bf16_multi_kernel, dim3(blocks + 1), dim3(256), 0, (hipStream_t)stream, a);
        WITW_CHECK_LAUNCH("bf16 pack_weights_multi");
    }
    return WITW_OK;
}

}  // extern "C"
