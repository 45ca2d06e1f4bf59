// Layers 0 and 2 of the FOV_DSM encoder in one kernel, bf16 inference (gfx950):
//   Conv2d(C<=8 -> 64, 3x3, pad 1) + ReLU  ->  Conv2d(64 -> 64, 3x3, pad 1) + ReLU + MaxPool2d(2,2)
// (features[0..4] of the VGG16 trunk, model/cvig_fov.py:256-260; cvig_semantic's 5-channel first conv, :301-303), straight
// from the NCHW fp32 image to the pooled NHWC bf16 map.
//
// Why fuse: run separately (conv_first.hip, conv3x3_bf16.hip) layer 0 is bound by its 64-channel output stream (1.07 GB per
// 128 images of 128x512) and layer 2, K = 64 only, spends 40 % of a workgroup's time outside the MFMA loop reading that stream
// back in: 290 + 713 us per encoder call, the worst-utilised 1 ms of the bf16 encoder. Here a persistent workgroup keeps the
// whole layer-2 filter (73.7 KB) in LDS for all of its tiles, recomputes layer 0 on the halo tile of each of its 8x32 layer-2
// output tiles from the raw pixels (layer 0 is 4.7 % of layer 2's FLOPs; with the halo 6.2 %) and never writes it out.
//
// Per tile (8 output rows x 32 columns x 64 channels, pooled 4 x 16 x 64):
//   A  the 12 x 36 raw pixels (one per thread, prefetched during the previous tile) -> LDS as bf16 (r,g,b,0) / 8-channel slots
//   B  layer 0 on the 10 x 34 positions of the layer-2 input tile: 11 M-tiles of 32 positions x 64 channels, MFMA with the
//      FILTER as the A operand so that a lane ends up with 4 consecutive channels of one position = one 8-byte LDS write into
//      the layer-2 operand image [channel group of 8][position]; positions outside the picture are layer 2's zero padding
//   C  layer 2: 4 K chunks x 9 taps of v_mfma_f32_32x32x16_bf16, M-tile = 2 rows x 16 columns (a 2x2 pooling window lies inside
//      one lane's registers), operands by ds_read_b128 from the two resident images, same accumulation order as
//      conv3x3_nhwc_bf16_kernel (chunk-major, tap-minor): bit-identical to the unfused launches
//   D  bias + ReLU + 2x2 max, through a wave-private 1 KB slab to 16-byte NHWC stores.
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int F2T = 512;                 // 8 waves
constexpr int TH2 = 8, TW2 = 32;         // layer-2 output tile
constexpr int AH = TH2 + 2, AW = TW2 + 2;          // layer-2 input tile (layer-0 outputs): 10 x 34 positions
constexpr int APITCH = 48;               // LDS row pitch of that tile in positions: a multiple of 16 keeps the two rows of an M-tile
                                         // (lanes 0-15 / 16-31) on disjoint banks under ds_read_b128's lane grouping
constexpr int APOS = AH * APITCH;        // 480 slots per channel group
constexpr int RH = TH2 + 4, RW = TW2 + 4;          // raw tile: 12 x 36 pixels
constexpr int NMT0 = (AH * AW + 31) / 32;          // 11 M-tiles of layer 0

struct First2Args {
    const float* x;           // NCHW fp32 [B,C,H,W]
    const u32x4* wf0;         // layer-0 filter image of conv3x3_first_bf16_kernel (witw_conv3x3_first_pack, round_bf16 = 1)
    const float* bias0;       // [64]
    const u32x4* wpk2;        // layer-2 filter, conv3x3_bf16 packing [4 chunks][9 taps][2 groups][64][8 bf16]
    const float* bias2;       // [64]
    unsigned short* y;        // NHWC bf16 [B,H/2,W/2,64]
    int B, C, H, W;
    int tiles_x, tiles_y;
    long long n_tiles;
    int circ;
};

template <int CW>
__global__ __launch_bounds__(F2T, 1) void conv_first2_bf16_kernel(First2Args p) {
    static_assert(CW == 4 || CW == 8, "4 or 8 bf16 per raw pixel");
    constexpr int NMF = (CW == 4) ? 3 : 5;
    typedef typename std::conditional<CW == 4, u32x2, u32x4>::type pix_t;
    __shared__ u32x4 a_s[8 * APOS];                 // 61,440 B: layer-0 output tile, [group of 8 channels][row][pitch 48]
    __shared__ u32x4 w_s[4 * 9 * 2 * 64];           // 73,728 B: layer-2 filter, resident for every tile of this workgroup
    __shared__ pix_t raw_s[RH * RW];                // 3.4 / 6.9 KB
    __shared__ u32x4 slab_s[8 * 64];                // 8 KB: one 1 KB output slab per wave
    __shared__ float b0_s[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hq = lane >> 5;
    const size_t plane = (size_t)p.H * p.W;
    const int Hy = p.H >> 1, Wy = p.W >> 1;

    // ---- once per workgroup: layer-2 filter -> LDS, layer-0 filter fragments and layer-2 bias -> registers
    for (int s = tid; s < 4 * 9 * 2 * 64; s += F2T) w_s[s] = p.wpk2[s];
    if (tid < 64) b0_s[tid] = p.bias0[tid];
    u32x4 aw[2][NMF];                               // layer-0 filter as the MFMA A operand: row = channel nt*32 + l31, k half = hq
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < NMF; ++i) aw[nt][i] = p.wf0[(i * 2 + hq) * 64 + nt * 32 + l31];
    float b2[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b2[nt] = p.bias2[nt * 32 + l31];

    // raw pixel of this thread (tile-relative), fetched one tile ahead
    const bool has_raw = tid < RH * RW;
    const int rr = tid / RW, rc = tid - rr * RW;
    float rv[CW];
    auto fetch_raw = [&](long long tile) {
#pragma unroll
        for (int ch = 0; ch < CW; ++ch) rv[ch] = 0.f;
        if (!has_raw || tile >= p.n_tiles) return;
        const int tiles_img = p.tiles_x * p.tiles_y;
        const int b = (int)(tile / tiles_img);
        const int rem = (int)(tile - (long long)b * tiles_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int gr = ty * TH2 - 2 + rr;
        int gc = tx * TW2 - 2 + rc;
        bool ok = gr >= 0 && gr < p.H;
        if (p.circ) {
            gc %= p.W;
            if (gc < 0) gc += p.W;
        } else {
            ok = ok && gc >= 0 && gc < p.W;
        }
        if (ok) {
            const float* src = p.x + (size_t)b * p.C * plane + (size_t)gr * p.W + gc;
#pragma unroll
            for (int ch = 0; ch < CW; ++ch)
                if (ch < p.C) rv[ch] = src[ch * plane];
        }
    };

    // layer-2 roles: wave = (row pair, column half); M-tile = 2 rows x 16 columns, lane l31 -> (row l31 >> 4, column l31 & 15)
    const int prow = wave >> 1, chalf = wave & 1;
    const int a_lane = ((2 * prow + (l31 >> 4)) * APITCH + 16 * chalf + (l31 & 15));      // + (kh * APITCH + kw) per tap

    fetch_raw(blockIdx.x);
    for (long long tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int tiles_img = p.tiles_x * p.tiles_y;
        const int b = (int)(tile / tiles_img);
        const int rem = (int)(tile - (long long)b * tiles_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * TH2, ox0 = tx * TW2;

        // ---- A: raw pixels -> LDS (bf16)
        if (has_raw) {
            __bf16 v[CW];
#pragma unroll
            for (int ch = 0; ch < CW; ++ch) v[ch] = (__bf16)rv[ch];
            raw_s[tid] = __builtin_bit_cast(pix_t, v);
        }
        __syncthreads();          // raw tile visible; every wave has left the previous tile's layer-2 loop (a_s is free)
        fetch_raw(tile + gridDim.x);

        // ---- B: layer 0 on the 10 x 34 positions, 11 M-tiles over the 8 waves
        for (int mt = wave; mt < NMT0; mt += 8) {
            const int q = mt * 32 + l31;                    // position index in the 10 x 34 tile
            const bool live = q < AH * AW;
            const int qr = live ? q / AW : 0, qc = live ? q - qr * AW : 0;
            const int rbase = qr * RW + qc;                 // raw pixel of tap (0,0)
            f32x16 acc0[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[nt][r] = 0.f;
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                u32x4 pv;
                if constexpr (CW == 4) {
                    const int tA = 4 * i + 2 * hq, tB = tA + 1;
                    u32x2 pa = raw_s[rbase + ((tA < 9) ? (tA / 3) * RW + tA % 3 : 0)];
                    u32x2 pb = raw_s[rbase + ((tB < 9) ? (tB / 3) * RW + tB % 3 : 0)];
                    if (tA >= 9) pa = (u32x2){0u, 0u};
                    if (tB >= 9) pb = (u32x2){0u, 0u};
                    pv = (u32x4){pa[0], pa[1], pb[0], pb[1]};
                } else {
                    const int t = 2 * i + hq;
                    pv = raw_s[rbase + ((t < 9) ? (t / 3) * RW + t % 3 : 0)];
                    if (t >= 9) pv = (u32x4){0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)     // D[channel][position]: filter rows x pixel columns
                    acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[nt][i]), __builtin_bit_cast(bf16x8, pv),
                                                                       acc0[nt], 0, 0, 0);
            }
            // position inside the picture? (outside: layer 2's zero padding, or the wrapped column under circular padding)
            const int gy = oy0 - 1 + qr;
            int gx = ox0 - 1 + qc;
            bool inside = live && gy >= 0 && gy < p.H;
            if (!p.circ) inside = inside && gx >= 0 && gx < p.W;
            if (live) {
                unsigned char* dst = reinterpret_cast<unsigned char*>(a_s) + ((size_t)(qr * APITCH + qc)) * 16 + 8 * hq;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {           // registers 4j..4j+3 = channels nt*32 + 8j + 4hq + {0..3}
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc0[nt][4 * j + e] + b0_s[nt * 32 + 8 * j + 4 * hq + e];
                            v = fmaxf(v, 0.f);
                            o[e] = (__bf16)(inside ? v : 0.f);
                        }
                        *reinterpret_cast<bf16x4*>(dst + (size_t)(nt * 4 + j) * (APOS * 16)) = o;
                    }
            }
        }
        __syncthreads();          // layer-2 input tile complete

        // ---- C: layer 2, 36 (chunk, tap) steps
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const u32x4 fa = a_s[(2 * kc + hq) * APOS + a_lane + kh * APITCH + kw];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const u32x4 fb = w_s[((kc * 9 + tap) * 2 + hq) * 64 + nt * 32 + l31];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                                     acc[nt], 0, 0, 0);
                }
            }
        }

        // ---- D: bias + ReLU + 2x2 max-pool -> slab [pooled column 0..7][64 channels] bf16 -> 16-byte stores
        // register r <-> pixel m = (r&3) + 8*(r>>2) + 4*hq of the M-tile, m = 16*row + column: the window of pooled column
        // jp = (r&3)/2 + 4*((r>>2)&1) + 2*hq is registers {r, r+1, r+8, r+9} (r&3 in {0,2}, r < 8)
        __bf16* slab = reinterpret_cast<__bf16*>(slab_s + wave * 64);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = (jj & 1) * 2 + (jj >> 1) * 4;
                const float m4 = fmaxf(fmaxf(acc[nt][r], acc[nt][r + 1]), fmaxf(acc[nt][r + 8], acc[nt][r + 9]));
                const int jp = (jj & 1) + 4 * (jj >> 1) + 2 * hq;
                slab[jp * 64 + nt * 32 + l31] = (__bf16)fmaxf(m4 + b2[nt], 0.f);
            }
        {
            const int jp = lane >> 3, c8 = (lane & 7) * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + jp * 64 + c8);
            const int py = (oy0 >> 1) + prow, px = (ox0 >> 1) + 8 * chalf + jp;
            if (py < Hy && px < Wy)
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.y + (((size_t)b * Hy + py) * Wy + px) * 64 + c8));
        }
    }
}

}  // namespace

extern "C" {

// x NCHW fp32 [B,C<=8,H,W] -> y NHWC bf16 [B,H/2,W/2,64] = MaxPool2(ReLU(conv2(ReLU(conv0(x))))), bf16 operands / fp32 accumulate.
// wf0 / bias0: witw_conv3x3_first_pack(round_bf16 = 1) image and bias of the first conv; wpk2 / bias2: witw_conv3x3_bf16_pack_weights
// image (64 -> 64) and bias of the second. Bit-identical to witw_conv3x3_first_fwd(out_bf16 = 1) followed by
// witw_conv3x3_bf16_fwd(relu, pool).
int witw_conv_first2_bf16_fwd(const float* x, const void* wf0, const float* bias0, const void* wpk2, const float* bias2, void* y,
                              int B, int C, int H, int W, int pad_circular, void* stream) {
    WITW_CHECK_ARG(x && wf0 && bias0 && wpk2 && bias2 && y, "conv_first2_bf16: null pointer");
    WITW_CHECK_ARG(B > 0 && C >= 1 && C <= 8 && H >= 2 && W >= 2, "conv_first2_bf16: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    First2Args a;
    a.x = x; a.wf0 = (const u32x4*)wf0; a.bias0 = bias0; a.wpk2 = (const u32x4*)wpk2; a.bias2 = bias2; a.y = (unsigned short*)y;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.tiles_x = cdiv(W, TW2); a.tiles_y = cdiv(H, TH2);
    a.n_tiles = (long long)B * a.tiles_x * a.tiles_y;
    a.circ = pad_circular;
    static int n_cu = 0;        // persistent workgroups, one per CU (150 KB of LDS each)
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const unsigned grid = (unsigned)(a.n_tiles < n_cu ? a.n_tiles : n_cu);
    if (C <= 4)
        hipLaunchKernelGGL(conv_first2_bf16_kernel<4>, dim3(grid), dim3(F2T), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(conv_first2_bf16_kernel<8>, dim3(grid), dim3(F2T), 0, (hipStream_t)stream, a);
    WITW_CHECK_LAUNCH("conv_first2_bf16");
    return WITW_OK;
}

}  // extern "C"
