// First encoder layer: Conv2d(C<=4 -> 64, 3x3, pad 1) + ReLU straight from the NCHW fp32 image (gfx950).
//
// Reference: features[0..1] of the VGG16 trunk (model/cvig_fov.py:256-260) fed by the Dataset's CHW tensors
// (:90-91). K = 27 is too short for the generic implicit-GEMM kernel, whose 8-channel K chunks would spend
// 72 MFMA k-steps on 27 products and which needs an NHWC8 copy of the input first. Here the two lane halves
// of v_mfma_f32_32x32x2_f32 take two different TAPS of the same pixel's (r,g,b,0) float4 instead of two
// channel groups, so a tile costs 5 tap-pairs x 3 channels = 15 k-steps (20 for 4 channels); the halo tile is gathered from the
// three NCHW planes directly. The layer is bound by its output stream (64 channels per pixel), which leaves
// through the same LDS-transposed 16-byte stores as the generic kernel, in fp32 or bf16.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int FT = 512;            // threads: 8 waves, 8 rows x 64 columns x 64 channels per workgroup
constexpr int FIW = 66, FIH = 10;

struct FirstArgs {
    const float* x;      // NCHW [B,C,H,W]
    const float* wf;     // packed [5 tap pairs][2][64][4]
    const float* bias;   // [64]
    void* y;             // NHWC [B,H,W,64] fp32 or bf16
    int B, C, H, W;
    int tiles_x, tiles_y;
    int circ, relu, out_bf16;
};

__global__ __launch_bounds__(FT, 2) void conv3x3_first_kernel(FirstArgs p) {
    __shared__ f32x4 smem[4096];                 // 64 KB: [0,660) input tile, [704,1344) weights; slabs alias all of it
    f32x4* in_s = smem;
    f32x4* w_s = smem + 704;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hq = lane >> 5;
    int bid = blockIdx.x;
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int b = bid / tiles_img;
    bid -= b * tiles_img;
    const int ty = bid / p.tiles_x, tx = bid - ty * p.tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 64;
    const size_t plane = (size_t)p.H * p.W;

    for (int s = tid; s < FIH * FIW; s += FT) {
        const int r = s / FIW, c = s - r * FIW;
        const int gr = oy0 - 1 + r;
        int gc = ox0 - 1 + c;
        bool ok = gr >= 0 && gr < p.H;
        if (p.circ) {
            gc %= p.W;
            if (gc < 0) gc += p.W;
        } else {
            ok = ok && gc >= 0 && gc < p.W;
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float* src = p.x + (size_t)b * p.C * plane + (size_t)gr * p.W + gc;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
                if (ch < p.C) v[ch] = (p.out_bf16 == 1) ? (float)(__bf16)src[ch * plane] : src[ch * plane];
        }
        in_s[s] = v;
    }
    for (int s = tid; s < 640; s += FT) w_s[s] = reinterpret_cast<const f32x4*>(p.wf)[s];
    __syncthreads();

    // wave -> 2 M-tiles (rows 2*(wave>>1), +1; column half wave&1) x 2 N-tiles
    const int row0 = 2 * (wave >> 1), col0 = 32 * (wave & 1);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tapA = 2 * i, tapB = (2 * i + 1 < 9) ? 2 * i + 1 : 0;     // tap 9 does not exist: zero weights
        const int offA = (tapA / 3) * FIW + tapA % 3, offB = (tapB / 3) * FIW + tapB % 3;
        const int off = hq ? offB : offA;
        f32x4 av[2], bw[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) av[mt] = in_s[(row0 + mt) * FIW + col0 + l31 + off];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bw[nt] = w_s[(i * 2 + hq) * 64 + nt * 32 + l31];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j == 3 && p.C < 4) break;       // RGB: the 4th channel of every slot is zero (uniform branch)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][j], bw[nt][j], acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();     // every wave is done with the input / weight images before the slabs overwrite them

    float bv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bv[nt] = p.bias[nt * 32 + l31];
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 64);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][nt][r] + bv[nt];
                if (p.relu) v = fmaxf(v, 0.f);
                slab[((r & 3) + 8 * (r >> 2) + 4 * hq) * 64 + nt * 32 + l31] = v;
            }
        const int yy = oy0 + row0 + mt;
        if (!p.out_bf16) {
            const int prow = lane >> 4, pc4 = (lane & 15) * 4;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int m = g * 4 + prow;
                const int xx = ox0 + col0 + m;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc4);
                if (yy < p.H && xx < p.W)      // streaming store: 2.1 GB written once per launch (0.66 -> 0.55 ms against a plain store)
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + (((size_t)b * p.H + yy) * p.W + xx) * 64 + pc4));
            }
        } else if (p.out_bf16 == 2) {
            // split-fp16 output of the fp16x3 path (conv3x3_f16x3.hip): exact fp32 arithmetic here, then per pixel and 8
            // channels 16 B of hi = fp16(v) followed by 16 B of lo = fp16(v - hi)
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            const int prow = lane >> 3, pc8 = (lane & 7) * 8;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = g * 8 + prow;
                const int xx = ox0 + col0 + m;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8 + 4);
                f16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (_Float16)v0[e];
                    hi[4 + e] = (_Float16)v1[e];
                    lo[e] = (_Float16)(v0[e] - (float)hi[e]);
                    lo[4 + e] = (_Float16)(v1[e] - (float)hi[4 + e]);
                }
                if (yy < p.H && xx < p.W) {
                    f16x8* dst = reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(p.y) + ((((size_t)b * p.H + yy) * p.W + xx) * 64 + pc8) * 2);
                    __builtin_nontemporal_store(hi, dst);
                    __builtin_nontemporal_store(lo, dst + 1);
                }
            }
        } else {
            const int prow = lane >> 3, pc8 = (lane & 7) * 8;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = g * 8 + prow;
                const int xx = ox0 + col0 + m;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8 + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (__bf16)v0[e];
                    o[4 + e] = (__bf16)v1[e];
                }
                if (yy < p.H && xx < p.W)
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + (((size_t)b * p.H + yy) * p.W + xx) * 64 + pc8) = o;
            }
        }
    }
}

// fp32 output, PERSISTENT form (round 4). The kernel above is bound by its 2.1 GB output stream (B = 128) but ran at half the rate a
// plain fill reaches on the same box (3.98 against 6.9 TB/s): a workgroup's life was input loads (their latency exposed: nothing
// else to do) -> 60 MFMAs per wave -> slabs -> stores, and its successor started from scratch. Here two workgroups per CU walk the
// tiles: the NEXT tile's pixels are requested before the MFMAs of the current one and written to the other half of a double-buffered
// input image at the end of the iteration (one barrier per tile), the slabs no longer alias the operands (4 KB per wave, one
// (M-tile, N-tile) at a time), so a workgroup's stores drain under its next tile's MFMAs and under the other workgroup's. Same
// products, same accumulation order, same bits as conv3x3_first_kernel.
constexpr int FIN = FIH * FIW;                 // 660 pixels of a tile's input image
constexpr int FPX = (FIN + FT - 1) / FT;       // input pixels per thread (2)

__global__ __launch_bounds__(FT, 2) void conv3x3_first_persist_kernel(FirstArgs p) {
    __shared__ f32x4 in2_s[2 * FIN];             // 21,120 B
    __shared__ f32x4 wp_s[640];                  // 10,240 B
    __shared__ float slab_s[8 * 16 * 64];        // 32,768 B: one 16 pixel x 64 channel slab per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hq = lane >> 5;
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int n_tiles = p.B * tiles_img;
    const size_t plane = (size_t)p.H * p.W;
    const unsigned img_bytes = (unsigned)p.C * (unsigned)plane * 4u;       // < 2^31 (launcher)
    constexpr unsigned OOR = 0xfffffff0u;

    // this thread's input pixels: tile-relative (row, column), the same for every tile
    int pr[FPX], pc[FPX];
#pragma unroll
    for (int k = 0; k < FPX; ++k) {
        const int s = tid + k * FT;
        pr[k] = s < FIN ? s / FIW : -100000;
        pc[k] = s < FIN ? s - (s / FIW) * FIW : 0;
    }
    float rv[FPX][4];
    auto fetch = [&](int tile) {                 // buffer loads: padding, missing channels and tiles past the end read zeros
        const bool any = tile < n_tiles;
        const int tl = any ? tile : 0;
        const int b = tl / tiles_img, rem = tl - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const unsigned char* img = reinterpret_cast<const unsigned char*>(p.x) + (size_t)b * img_bytes;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, any ? img_bytes : 0u, 0x00020000);
#pragma unroll
        for (int k = 0; k < FPX; ++k) {
            const int gr = ty * 8 - 1 + pr[k];
            int gc = tx * 64 - 1 + pc[k];
            bool ok = gr >= 0 && gr < p.H;
            if (p.circ) {                        // gc in [-1, W + 64]: W >= 66 (launcher) -> one wrap
                gc += gc < 0 ? p.W : 0;
                gc -= gc >= p.W ? p.W : 0;
            } else {
                ok = ok && gc >= 0 && gc < p.W;
            }
            const unsigned off = ok ? (unsigned)(gr * p.W + gc) * 4u : OOR;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)       // plane ch rides in soffset, which the range check does NOT cover (it sees voffset only):
                // a missing plane (ch >= C, wave-uniform) is made out of range through the voffset, so that it reads zeros, not the next image
                rv[k][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ch < p.C ? off : OOR,
                                                                                           (unsigned)ch * (unsigned)plane * 4u, 0));
        }
    };
    auto to_lds = [&](int buf) {
#pragma unroll
        for (int k = 0; k < FPX; ++k)
            if (tid + k * FT < FIN) in2_s[buf * FIN + tid + k * FT] = (f32x4){rv[k][0], rv[k][1], rv[k][2], rv[k][3]};
    };

    for (int s = tid; s < 640; s += FT) wp_s[s] = reinterpret_cast<const f32x4*>(p.wf)[s];
    float bv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bv[nt] = p.bias[nt * 32 + l31];
    const int row0 = 2 * (wave >> 1), col0 = 32 * (wave & 1);
    float* slab = slab_s + wave * (16 * 64);

    fetch(blockIdx.x);
    to_lds(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // everything loaded so far has landed (keeps vmcnt(0) out of the loop body)
    int it = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++it) {
        const int b = tile / tiles_img, rem = tile - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * 8, ox0 = tx * 64;
        const f32x4* in_s = in2_s + (it & 1) * FIN;
        __syncthreads();                         // this tile's input image is complete; the other half is free
        fetch(tile + (int)gridDim.x);

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int tapA = 2 * i, tapB = (2 * i + 1 < 9) ? 2 * i + 1 : 0;     // tap 9 does not exist: zero weights
            const int offA = (tapA / 3) * FIW + tapA % 3, offB = (tapB / 3) * FIW + tapB % 3;
            const int off = hq ? offB : offA;
            f32x4 av[2], bw[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) av[mt] = in_s[(row0 + mt) * FIW + col0 + l31 + off];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bw[nt] = wp_s[(i * 2 + hq) * 64 + nt * 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j == 3 && p.C < 4) break;       // RGB: the 4th channel of every slot is zero (uniform branch)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][j], bw[nt][j], acc[mt][nt], 0, 0, 0);
            }
        }
        // epilogue: half an M-tile (16 pixels x all 64 channels = registers 8h .. 8h+7 of both N-tiles) at a time through the wave's
        // 4 KB slab -> 16-byte stores of 4 pixels x 256 bytes = 1 KB contiguous per wave instruction, as conv3x3_first_kernel's; a
        // wave's LDS operations execute in order, so a round's writes need no wait for the previous round's reads
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int yy = oy0 + row0 + mt;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = 8 * h + q;            // pixel (r&3) + 8*(r>>2) + 4*hq = 16h + (q&3) + 8*(q>>2) + 4*hq
                        float v = acc[mt][nt][r] + bv[nt];
                        if (p.relu) v = fmaxf(v, 0.f);
                        slab[((q & 3) + 8 * (q >> 2) + 4 * hq) * 64 + nt * 32 + l31] = v;
                    }
                __builtin_amdgcn_wave_barrier();
                const int prow = lane >> 4, pc4 = (lane & 15) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = g * 4 + prow;             // pixel 0..15 of the half
                    const int xx = ox0 + col0 + 16 * h + m;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc4);
                    if (yy < p.H && xx < p.W)
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + (((size_t)b * p.H + yy) * p.W + xx) * 64 + pc4));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        to_lds((it + 1) & 1);                    // the next tile's pixels (requested before the MFMAs) -> the free half
    }
}

// bf16 form (the bf16 path's first layer: bf16-rounded operands, fp32 accumulate, bf16 NHWC out): with a pixel's
// (r,g,b,0) held as four bf16 (8 bytes), one v_mfma_f32_32x32x16_bf16 operand = 8 k values = TWO taps of a pixel, so the
// 9 taps take 3 MFMAs per tile (lane half h of MFMA i holds taps 4i+2h, 4i+2h+1; taps 9-11 do not exist: zeros)
// instead of 15 fp32 k-steps of twice the cycles: the layer is then purely bound by its 64-channel output stream.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// CW = bf16 values stored per pixel: 4 (C <= 4, 8 bytes: an MFMA operand = two taps, 3 MFMAs per tile) or 8 (cvig_semantic's
// five channels, C <= 8, 16 bytes: an operand = one tap, lane half h of MFMA i holds tap 2i+h, 5 MFMAs per tile).
template <int CW>
__global__ __launch_bounds__(FT, 2) void conv3x3_first_bf16_kernel(FirstArgs p) {
    static_assert(CW == 4 || CW == 8, "4 or 8 bf16 per pixel");
    constexpr int NMF = (CW == 4) ? 3 : 5;                 // MFMAs per (M-tile, N-tile)
    constexpr int IN_F4 = (CW == 4) ? 352 : 672;          // f32x4 slots reserved for the input tile (660 pixels)
    typedef typename std::conditional<CW == 4, u32x2, u32x4v>::type pix_t;
    __shared__ f32x4 smem[4096];                 // 64 KB: input tile, weights (NMF x 128 x 16 B); slabs alias all of it
    pix_t* in_s = reinterpret_cast<pix_t*>(smem);
    u32x4v* w_s = reinterpret_cast<u32x4v*>(smem + IN_F4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hq = lane >> 5;
    int bid = blockIdx.x;
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int b = bid / tiles_img;
    bid -= b * tiles_img;
    const int ty = bid / p.tiles_x, tx = bid - ty * p.tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 64;
    const size_t plane = (size_t)p.H * p.W;

    for (int s = tid; s < FIH * FIW; s += FT) {
        const int r = s / FIW, c = s - r * FIW;
        const int gr = oy0 - 1 + r;
        int gc = ox0 - 1 + c;
        bool ok = gr >= 0 && gr < p.H;
        if (p.circ) {
            gc %= p.W;
            if (gc < 0) gc += p.W;
        } else {
            ok = ok && gc >= 0 && gc < p.W;
        }
        __bf16 v[CW];
#pragma unroll
        for (int ch = 0; ch < CW; ++ch) v[ch] = (__bf16)0.f;
        if (ok) {
            const float* src = p.x + (size_t)b * p.C * plane + (size_t)gr * p.W + gc;
#pragma unroll
            for (int ch = 0; ch < CW; ++ch)
                if (ch < p.C) v[ch] = (__bf16)src[ch * plane];
        }
        in_s[s] = __builtin_bit_cast(pix_t, v);
    }
    for (int s = tid; s < NMF * 128; s += FT) w_s[s] = reinterpret_cast<const u32x4v*>(p.wf)[s];
    __syncthreads();

    const int row0 = 2 * (wave >> 1), col0 = 32 * (wave & 1);
    float bv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bv[nt] = p.bias[nt * 32 + l31];
    f32x16 acc[2][2];          // started from the bias (conv_first2_bf16.hip does the same: the two stay bit-identical)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = bv[c];
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
        u32x4v bw[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bw[nt] = w_s[(i * 2 + hq) * 64 + nt * 32 + l31];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int base = (row0 + mt) * FIW + col0 + l31;
            u32x4v av;
            if constexpr (CW == 4) {
                // this lane's two taps of MFMA i; a tap past 8 reads tap 0's pixel and is zeroed
                const int tA = 4 * i + 2 * hq, tB = tA + 1;
                const int offA = (tA < 9) ? (tA / 3) * FIW + tA % 3 : 0;
                const int offB = (tB < 9) ? (tB / 3) * FIW + tB % 3 : 0;
                u32x2 pa = in_s[base + offA], pb = in_s[base + offB];
                if (tA >= 9) pa = (u32x2){0u, 0u};
                if (tB >= 9) pb = (u32x2){0u, 0u};
                av = (u32x4v){pa[0], pa[1], pb[0], pb[1]};
            } else {
                const int t = 2 * i + hq;                  // tap 9 does not exist: zeros (its filter slot is zero as well)
                av = in_s[base + ((t < 9) ? (t / 3) * FIW + t % 3 : 0)];
                if (t >= 9) av = (u32x4v){0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bw[nt]),
                                                                     acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();     // every wave is done with the input / weight images before the slabs overwrite them

    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    const int prow = lane >> 3, pc8 = (lane & 7) * 8;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][nt][r];
                if (p.relu) v = fmaxf(v, 0.f);
                slab[((r & 3) + 8 * (r >> 2) + 4 * hq) * 64 + nt * 32 + l31] = v;
            }
        const int yy = oy0 + row0 + mt;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = g * 8 + prow;
            const int xx = ox0 + col0 + m;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8 + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (__bf16)v0[e];
                o[4 + e] = (__bf16)v1[e];
            }
            if (yy < p.H && xx < p.W)
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + (((size_t)b * p.H + yy) * p.W + xx) * 64 + pc8));
        }
    }
}

// bf16 filter image of conv3x3_first_bf16_kernel: slot [i][h][n] (16 B) = (w[n][0..3][tap 4i+2h], w[n][0..3][tap 4i+2h+1])
// as bf16, zeros for taps >= 9 and channels >= C; 384 slots = 1536 floats of the 2560-float buffer.
// C > 4 (the 8-values-per-pixel form): slot [i][h][n] = w[n][0..7][tap 2i+h], 640 slots = the whole 2560-float buffer.
__global__ void pack_first_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int slots = (C <= 4) ? 384 : 640;
    if (idx >= slots) return;
    const int n = idx % 64, h = (idx / 64) % 2, i = idx / 128;
    __bf16 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int tap = (C <= 4) ? 4 * i + 2 * h + (e >> 2) : 2 * i + h;
        const int ch = (C <= 4) ? (e & 3) : e;
        v[e] = (__bf16)((tap < 9 && ch < C) ? w[((size_t)n * C + ch) * 9 + tap] : 0.f);
    }
    reinterpret_cast<bf16x8*>(wf)[idx] = __builtin_bit_cast(bf16x8, v);
}

// wf[i][h][n][0..3] = (w[n][0][tap], w[n][1][tap], w[n][2][tap], w[n][3][tap]) with tap = 2i+h (zeros for tap 9
// and for channels >= C); round_bf16 != 0 rounds the weights to bf16 first (the bf16 path's filters).
__global__ void pack_first_kernel(const float* __restrict__ w, float* __restrict__ wf, int C, int round_bf16) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 640) return;
    const int n = idx % 64, h = (idx / 64) % 2, i = idx / 128;
    const int tap = 2 * i + h;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tap < 9)
        for (int ch = 0; ch < C && ch < 4; ++ch) {
            float f = w[((size_t)n * C + ch) * 9 + tap];
            if (round_bf16) f = (float)(__bf16)f;
            v[ch] = f;
        }
    reinterpret_cast<f32x4*>(wf)[idx] = v;
}

int first_persistent() {     // WITW_FIRST_PERSIST=0: the one-tile-per-workgroup kernel everywhere (A/B)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("WITW_FIRST_PERSIST");
        v = e ? atoi(e) != 0 : 1;
    }
    return v;
}

}  // namespace

extern "C" {

// w: torch layout [64][C][3][3], C <= 4 (bf16 image: C <= 8) -> wf: 2560 floats
int witw_conv3x3_first_pack(const float* w, float* wf, int C, int round_bf16, void* stream) {
    WITW_CHECK_ARG(w && wf, "conv3x3_first_pack: null pointer");
    WITW_CHECK_ARG(C >= 1 && C <= (round_bf16 ? 8 : 4), "conv3x3_first_pack: C=%d outside [1,%d]", C, round_bf16 ? 8 : 4);
    if (round_bf16)     // the bf16 kernel's own filter image (see pack_first_bf16_kernel)
        hipLaunchKernelGGL(pack_first_bf16_kernel, dim3(3), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wf, C);
    else
        hipLaunchKernelGGL(pack_first_kernel, dim3(3), dim3(256), 0, (hipStream_t)stream, w, wf, C, round_bf16);
    WITW_CHECK_LAUNCH("conv3x3_first_pack");
    return WITW_OK;
}

// x NCHW fp32 [B,C,H,W] (C <= 4; C <= 8 with out_bf16 = 1) -> y NHWC [B,H,W,64] (fp32, or bf16 if out_bf16; with out_bf16 the INPUT is
// rounded to bf16 on load so that the arithmetic equals the bf16 path's: bf16 operands, fp32 accumulate, and wf
// must come from witw_conv3x3_first_pack(round_bf16 = 1), which writes the bf16 kernel's filter image). out_bf16 = 2: exact
// fp32 arithmetic (wf packed with round_bf16 = 0) and a split-fp16 output [B,H,W,8,2,8] for the fp16x3 path.
int witw_conv3x3_first_fwd(const float* x, const float* wf, const float* bias, void* y, int B, int C, int H, int W,
                           int pad_circular, int relu, int out_bf16, void* stream) {
    WITW_CHECK_ARG(x && wf && bias && y, "conv3x3_first_fwd: null pointer");
    WITW_CHECK_ARG(B > 0 && C >= 1 && C <= (out_bf16 == 1 ? 8 : 4) && H > 0 && W > 0, "conv3x3_first_fwd: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    FirstArgs a;
    a.x = x; a.wf = wf; a.bias = bias; a.y = y;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.tiles_x = cdiv(W, 64); a.tiles_y = cdiv(H, 8);
    a.circ = pad_circular; a.relu = relu; a.out_bf16 = out_bf16;
    const long long grid = (long long)B * a.tiles_x * a.tiles_y;
    WITW_CHECK_ARG(grid <= 0x7fffffffLL, "conv3x3_first_fwd: grid too large");
    if (out_bf16 == 1 && C > 4) {
        hipLaunchKernelGGL(conv3x3_first_bf16_kernel<8>, dim3((unsigned)grid), dim3(FT), 0, (hipStream_t)stream, a);
        witw_note_variant("conv3x3_first_bf16_kernel<8>");
    } else if (out_bf16 == 1) {
        hipLaunchKernelGGL(conv3x3_first_bf16_kernel<4>, dim3((unsigned)grid), dim3(FT), 0, (hipStream_t)stream, a);
        witw_note_variant("conv3x3_first_bf16_kernel<4>");
    } else if (out_bf16 == 0 && W >= 66 && (unsigned long long)C * H * W * 4 < 0x80000000ull && grid >= 4LL * witw_cu_count() && first_persistent()) {
        const unsigned g2 = 2u * (unsigned)witw_cu_count();      // two persistent workgroups per CU
        hipLaunchKernelGGL(conv3x3_first_persist_kernel, dim3(g2), dim3(FT), 0, (hipStream_t)stream, a);
        witw_note_variant("conv3x3_first_persist_kernel");
    } else {
        hipLaunchKernelGGL(conv3x3_first_kernel, dim3((unsigned)grid), dim3(FT), 0, (hipStream_t)stream, a);
        witw_note_variant("conv3x3_first_kernel");
    }
    WITW_CHECK_LAUNCH("conv3x3_first_fwd");
    return WITW_OK;
}

}  // extern "C"
