// Device data path of the reference's host transforms (gfx950, HBM-bound byte/pixel work):
//   Resize              model/cvig_fov.py:117-134 (torchvision F.resize on float CHW = bilinear,
//                       align_corners=False, no antialias under the pinned torchvision 0.9.1)
//   ImageNormalization  model/cvig_fov.py:137-149  ((x/255 - mean)/std; semantic variant
//                       model/cvig_semantic.py:172-176 divides only channels 0-2 by 255)
//   PolarTransform      model/cvig_fov.py:186-209 + bilinear_interpolate :156-183
//   SyncedRotation      model/cvig_baseline.py:131-144 (torchvision F.rotate on a float CHW tensor = affine grid +
//                       grid_sample(nearest, zeros, align_corners=False) under torchvision 0.9.1)
// Compiled with -ffp-contract=off so products and sums round exactly like the reference's
// separate elementwise torch ops (polar: wa*Ia + wb*Ib + wc*Ic + wd*Id, left to right).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace {

struct NormArgs {
    float mean[8];
    float stdv[8];
    float rstd[8];  // RN(1/stdv[c]) where the three-operation division below is exact, else 0 (-> a true division)
    int n_div255;   // channels [0,n_div255) are divided by 255 first
    int enabled;
};

// a / b, correctly rounded, without the division sequence: with y = RN(1/b), q0 = RN(a y), r = a - b q0 (exact in one fma),
// q1 = RN(q0 + r y) is the IEEE quotient (Markstein) for every finite a whose quotient is a normal number, unless b's
// significand is all ones; fill_norm() hands out y only for divisors in [2^-10, 2^10] that pass that test, and
// tools/debug/markstein.py compares 1.7e8 dividends per divisor of the two models with the true division: no difference.
// y == 0 selects the true division.
__device__ __forceinline__ float div_exact(float a, float b, float y) {
    if (y == 0.f) return a / b;
    const float q0 = a * y;
    const float r = __fmaf_rn(-b, q0, a);
    return __fmaf_rn(r, y, q0);
}
constexpr float RCP_255 = 0x1.010102p-8f;      // RN(1/255)


// One thread per output pixel (b, y, x); loops over channels. Source index / lambda follow
// ATen's area_pixel_compute_source_index (align_corners=False): src = scale*(dst+0.5)-0.5, <0 -> 0.
// CT > 0: the channel count as a compile-time constant -- the loop is unrolled and all 4 x CT taps are requested before the first is
// used (the run-time loop keeps 4 loads in flight per thread: PMC showed the waves 73 % of their cycles in waits); CT = 0: any C.
template <int CT>
__global__ void resize_bilinear_norm_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int Hi, int Wi,
                                            int Ho, int Wo, float sh, float sw, NormArgs na) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo;
    if (idx >= total) return;
    const int ox = idx % Wo;
    size_t t = idx / Wo;
    const int oy = t % Ho;
    const int b = (int)(t / Ho);
    // ATen's CPU build contracts scale*(dst+0.5)-0.5 into one fma; follow it (the last bit of the
    // source coordinate moves lambda by ~1e-5, i.e. up to 4e-3 on 0..255 pixel values)
    float fy = __fmaf_rn(sh, oy + 0.5f, -0.5f);
    if (fy < 0.f) fy = 0.f;
    float fx = __fmaf_rn(sw, ox + 0.5f, -0.5f);
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < Hi - 1) ? 1 : 0), x1 = x0 + ((x0 < Wi - 1) ? 1 : 0);
    const float ly1 = fy - y0, lx1 = fx - x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    if (CT > 0) {
        float t00[CT > 0 ? CT : 1], t01[CT > 0 ? CT : 1], t10[CT > 0 ? CT : 1], t11[CT > 0 ? CT : 1];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float* p = x + ((size_t)b * CT + c) * Hi * Wi;
            t00[c] = p[(size_t)y0 * Wi + x0]; t01[c] = p[(size_t)y0 * Wi + x1];
            t10[c] = p[(size_t)y1 * Wi + x0]; t11[c] = p[(size_t)y1 * Wi + x1];
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float top = lx0 * t00[c] + lx1 * t01[c];
            const float bot = lx0 * t10[c] + lx1 * t11[c];
            float v = ly0 * top + ly1 * bot;
            if (na.enabled) {
                if (c < na.n_div255) v = div_exact(v, 255.f, RCP_255);
                v = div_exact(v - na.mean[c], na.stdv[c], na.rstd[c]);
            }
            y[(((size_t)b * CT + c) * Ho + oy) * Wo + ox] = v;
        }
        return;
    }
    for (int c = 0; c < C; ++c) {
        const float* p = x + ((size_t)b * C + c) * Hi * Wi;
        const float top = lx0 * p[(size_t)y0 * Wi + x0] + lx1 * p[(size_t)y0 * Wi + x1];
        const float bot = lx0 * p[(size_t)y1 * Wi + x0] + lx1 * p[(size_t)y1 * Wi + x1];
        float v = ly0 * top + ly1 * bot;
        if (na.enabled) {
            if (c < na.n_div255) v = div_exact(v, 255.f, RCP_255);
            v = div_exact(v - na.mean[c], na.stdv[c], na.rstd[c]);
        }
        y[(((size_t)b * C + c) * Ho + oy) * Wo + ox] = v;
    }
}

// The same resize (+ normalisation) for a BATCH of images of different sizes in one launch: image b is described by
// desc[b] = {address, H, W, start, pixel layout}; KIND 0 = float32 planar CHW (what ImagePairDataset returns, reference
// model/cvig_fov.py:88-95), KIND 1 = uint8 interleaved HWC straight from the decoder (a quarter of the bytes over PCIe),
// converted to float exactly (0..255). The panorama branch of Resize (:118-128: resize to Wfull columns, then keep Wo of them
// from a per-sample start, wrapping) is the column map ox -> (ox + start) mod Wfull. Same arithmetic, operation for
// operation, as resize_bilinear_norm_kernel.
struct ImgDesc {
    unsigned long long ptr;
    long long H, W, start, cs;      // cs: channels stored per pixel (KIND 1) -- the first C are used
};

template <int KIND, int CT>
__global__ void resize_batched_kernel(const ImgDesc* __restrict__ desc, float* __restrict__ y, int B, int C, int Ho, int Wo,
                                      int Wfull, NormArgs na) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo;
    if (idx >= total) return;
    const int ox = idx % Wo;
    size_t t = idx / Wo;
    const int oy = t % Ho;
    const int b = (int)(t / Ho);
    const ImgDesc d = desc[b];
    const int Hi = (int)d.H, Wi = (int)d.W;
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wfull;
    int cx = ox + (int)d.start;
    if (cx >= Wfull) cx -= Wfull;
    float fy = __fmaf_rn(sh, oy + 0.5f, -0.5f);
    if (fy < 0.f) fy = 0.f;
    float fx = __fmaf_rn(sw, cx + 0.5f, -0.5f);
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < Hi - 1) ? 1 : 0), x1 = x0 + ((x0 < Wi - 1) ? 1 : 0);
    const float ly1 = fy - y0, lx1 = fx - x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    if (CT > 0) {      // compile-time channel count: all 4 x CT taps in flight before the first use (see resize_bilinear_norm_kernel)
        float t00[CT > 0 ? CT : 1], t01[CT > 0 ? CT : 1], t10[CT > 0 ? CT : 1], t11[CT > 0 ? CT : 1];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (KIND == 0) {
                const float* p = reinterpret_cast<const float*>(d.ptr) + (size_t)c * Hi * Wi;
                t00[c] = p[(size_t)y0 * Wi + x0]; t01[c] = p[(size_t)y0 * Wi + x1];
                t10[c] = p[(size_t)y1 * Wi + x0]; t11[c] = p[(size_t)y1 * Wi + x1];
            } else {
                const unsigned char* p = reinterpret_cast<const unsigned char*>(d.ptr) + c;
                const size_t cs = (size_t)d.cs;
                t00[c] = (float)p[((size_t)y0 * Wi + x0) * cs]; t01[c] = (float)p[((size_t)y0 * Wi + x1) * cs];
                t10[c] = (float)p[((size_t)y1 * Wi + x0) * cs]; t11[c] = (float)p[((size_t)y1 * Wi + x1) * cs];
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float top = lx0 * t00[c] + lx1 * t01[c];
            const float bot = lx0 * t10[c] + lx1 * t11[c];
            float v = ly0 * top + ly1 * bot;
            if (na.enabled) {
                if (c < na.n_div255) v = div_exact(v, 255.f, RCP_255);
                v = div_exact(v - na.mean[c], na.stdv[c], na.rstd[c]);
            }
            y[(((size_t)b * CT + c) * Ho + oy) * Wo + ox] = v;
        }
        return;
    }
    for (int c = 0; c < C; ++c) {
        float p00, p01, p10, p11;
        if (KIND == 0) {
            const float* p = reinterpret_cast<const float*>(d.ptr) + (size_t)c * Hi * Wi;
            p00 = p[(size_t)y0 * Wi + x0]; p01 = p[(size_t)y0 * Wi + x1];
            p10 = p[(size_t)y1 * Wi + x0]; p11 = p[(size_t)y1 * Wi + x1];
        } else {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(d.ptr) + c;
            const size_t cs = (size_t)d.cs;
            p00 = (float)p[((size_t)y0 * Wi + x0) * cs]; p01 = (float)p[((size_t)y0 * Wi + x1) * cs];
            p10 = (float)p[((size_t)y1 * Wi + x0) * cs]; p11 = (float)p[((size_t)y1 * Wi + x1) * cs];
        }
        const float top = lx0 * p00 + lx1 * p01;
        const float bot = lx0 * p10 + lx1 * p11;
        float v = ly0 * top + ly1 * bot;
        if (na.enabled) {
            if (c < na.n_div255) v = div_exact(v, 255.f, RCP_255);
            v = div_exact(v - na.mean[c], na.stdv[c], na.rstd[c]);
        }
        y[(((size_t)b * C + c) * Ho + oy) * Wo + ox] = v;
    }
}

// ---- Resize -> ImageNormalization -> PolarTransform of the overhead image in ONE pass (model/cvig_fov.py:117-209, the
// overhead branch of Compose[Resize, ImageNormalization, PolarTransform]): the size x size resized + normalised image is never
// written. The output is cut into tiles of RH radii (rows) x AW angles (columns), RH * AW <= 256; a WAVE owns one tile for a run
// of (image, channel) planes and never synchronises with another wave. Its sampling-table entries (4 outputs per lane) are
// loaded ONCE into registers, the taps as offsets inside the tile's bounding box of the resized image; per plane the wave
// (1) computes the resized + normalised pixels of that box from the raw image into its own LDS area -- operation for
// operation resize_batched_kernel's arithmetic, the separable row / column terms (source offsets, second-tap weight) from
// small LDS tables that are rebuilt only when the source size changes --, (2) gathers the four taps of every output from LDS
// and combines them exactly as polar_kernel does. Bit-identical to the three launches. Waves are dealt so that the tiles of
// one run of planes share an XCD: the boxes of neighbouring tiles overlap and the raw image crosses the fabric once.
struct PolarTile {
    int bx0, by0, bw, bh;     // bounding box in the size x size plane of every tap of the tile
    int r0, c0, rh, aw;       // output rows [r0, r0+rh) x columns [c0, c0+aw)
};

struct PolarRawArgs {
    const void* src;          // KIND 0: fp32 planar [B,C,Hi,Wi]; unused when desc != nullptr
    const ImgDesc* desc;      // per-image {address, H, W, -, channels per stored pixel} or nullptr (uniform fp32 batch)
    float* y;                 // [B,C,Ho,Wo]
    const int4* taps;
    const float4* wts;
    const PolarTile* tile;
    int B, C, Hi, Wi, size, Ho, Wo, n_tile, planes_per_wave, n_group, box_stride;
    NormArgs na;
};

constexpr int PR_WAVES = 4;     // independent waves per workgroup
constexpr int PR_OUT = 4;       // outputs per lane: rh * aw <= 64 * PR_OUT
constexpr int PR_MAXE = 64;     // widest / tallest tile box

// LDS traffic of ONE wave is processed in issue order, so a hand-over between its lanes needs no barrier: only the compiler has
// to be kept from moving LDS accesses across this point, and the LDS counter drained. (A fence builtin would also wait for
// vmcnt, i.e. for the previous plane's output STORES: ~2 us per plane with nothing to overlap them.)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

#ifndef PR_OCC
#define PR_OCC 4
#endif
typedef const __attribute__((address_space(1))) unsigned char* gmem_u8;
typedef const __attribute__((address_space(1))) float* gmem_f32;

#ifndef PR_NPL_N
#define PR_NPL_N 1
#endif
// planes of equal source size a wave takes through one pass (shared index / address arithmetic). Measured 1 / 2 / 3: the same
// 165 us at 128 x 3 x 512^2 (the pass is bound by the issue of its LDS + vector instructions, not by their count per plane);
// 1 keeps the register count (100) and the LDS area per wave (4.5 KB) lowest.
constexpr int PR_NPL = PR_NPL_N;

struct PolarPlane {             // per-plane scalars (SGPRs)
    gmem_u8 base;
    float* y;
    float mean, stdv, rstd;
    bool div255;
};

// NPL planes of one tile: (1) their boxes of the resized + normalised image into LDS, (2) the tile's outputs from LDS.
// HALF: the source is exactly twice the resized image in both directions (BASELINE's 512 x 512 -> 256 x 256): source pixel = 2 x
// resized pixel, both weights 0.5 -- the same operations on the same operands as the general path (so the same bits), but no
// term tables, and the two taps of a row arrive as one 8-byte load.
template <int KIND, int NPL, bool HALF>
__device__ __forceinline__ void polar_planes(const PolarPlane (&pp)[PR_NPL], bool norm, int lane, float* bp, int box_stride, const unsigned* cx0,
                                             const unsigned* cx1, const float* clx1, const unsigned* ry0, const unsigned* ry1,
                                             const float* rly1, int bw, int n_box, float inv_bw, const unsigned (&o01)[PR_OUT],
                                             const unsigned (&o23)[PR_OUT], const int (&dst)[PR_OUT], const float4 (&wt)[PR_OUT],
                                             int bx0, int by0, unsigned row_bytes, unsigned px_bytes) {
    // box pixels in batches of MLP per lane (their 4 * NPL loads each in flight together): batches of 4 while 256 pixels remain, then
    // batches of 2 -- a box of 375 pixels (the mean of the 16 x 16 tiles) costs 384 pixel slots, not 768
#ifndef WITW_PR_NOP1
    auto batch = [&](auto mlp_c, int q0) {
        constexpr int MLP = decltype(mlp_c)::value;
        float p00[NPL][MLP], p01[NPL][MLP], p10[NPL][MLP], p11[NPL][MLP], lx1[MLP], ly1[MLP];
#pragma unroll
        for (int k = 0; k < MLP; ++k) {
            int q = q0 + 64 * k;
            if (q >= n_box) q = n_box - 1;      // the tail repeats the last pixel (not stored)
            const int ry = (int)(((float)q + 0.5f) * inv_bw), rx = q - ry * bw;
            unsigned a00, a01, a10, a11;      // byte offsets of the four taps, the same in every plane
            if (HALF) {
                a00 = (unsigned)(2 * (by0 + ry)) * row_bytes + (unsigned)(2 * (bx0 + rx)) * px_bytes;
                a01 = a00 + px_bytes; a10 = a00 + row_bytes; a11 = a10 + px_bytes;
                ly1[k] = 0.5f; lx1[k] = 0.5f;
            } else {
                const unsigned r0 = ry0[ry], r1 = ry1[ry], x0 = cx0[rx], x1 = cx1[rx];
                ly1[k] = rly1[ry];
                lx1[k] = clx1[rx];
                a00 = r0 + x0; a01 = r0 + x1; a10 = r1 + x0; a11 = r1 + x1;
            }
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (KIND == 0 && HALF) {          // neighbouring taps: one 8-byte load per source row
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef const __attribute__((address_space(1))) f32x2* gmem_f32x2;
                    const f32x2 t0 = *reinterpret_cast<gmem_f32x2>(pp[j].base + a00), t1 = *reinterpret_cast<gmem_f32x2>(pp[j].base + a10);
                    p00[j][k] = t0[0]; p01[j][k] = t0[1]; p10[j][k] = t1[0]; p11[j][k] = t1[1];
                } else if (KIND == 0) {
                    p00[j][k] = *reinterpret_cast<gmem_f32>(pp[j].base + a00); p01[j][k] = *reinterpret_cast<gmem_f32>(pp[j].base + a01);
                    p10[j][k] = *reinterpret_cast<gmem_f32>(pp[j].base + a10); p11[j][k] = *reinterpret_cast<gmem_f32>(pp[j].base + a11);
                } else {
                    p00[j][k] = (float)pp[j].base[a00]; p01[j][k] = (float)pp[j].base[a01];
                    p10[j][k] = (float)pp[j].base[a10]; p11[j][k] = (float)pp[j].base[a11];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MLP; ++k) {
            const int q = q0 + 64 * k;
            const float ly0 = 1.f - ly1[k], lx0 = 1.f - lx1[k];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const float top = lx0 * p00[j][k] + lx1[k] * p01[j][k];
                const float bot = lx0 * p10[j][k] + lx1[k] * p11[j][k];
                float v = ly0 * top + ly1[k] * bot;
                if (norm) {
                    if (pp[j].div255) v = div_exact(v, 255.f, RCP_255);
                    v = div_exact(v - pp[j].mean, pp[j].stdv, pp[j].rstd);
                }
                if (q < n_box) bp[j * box_stride + q] = v;
            }
        }
    };
    {
        constexpr int MB = NPL == 1 ? 4 : 2;      // the big batch
        int base = 0;
        for (; base + 64 * MB <= n_box; base += 64 * MB) batch(std::integral_constant<int, MB>(), base + lane);
        for (; base < n_box; base += 64 * (MB / 2)) batch(std::integral_constant<int, MB / 2>(), base + lane);
    }
#endif
    // Every load of the plane has been consumed (waited for) above; said once more with the builtin, for the compiler's bookkeeping:
    // the batches sit behind run-time loop bounds and a predicated tail, so it kept some of their destination registers "pending"
    // on some path and put a vmcnt(0) in front of each of the gathers below that reuses such a register -- i.e. behind the output
    // store issued just before: four serialised store round trips per plane (148 -> 138 us at 128 x 3 x 512^2).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < PR_OUT; ++i) {
#ifdef WITW_PR_NOP2
        if (dst[i] == -7) pp[0].y[0] = wt[i].x + (float)(o01[i] + o23[i]);
#else
        if (dst[i] >= 0) {
            const float4 ww = wt[i];
            const unsigned a = o01[i] & 0xffffu, b = o01[i] >> 16, c = o23[i] & 0xffffu, d = o23[i] >> 16;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const float* bj = bp + j * box_stride;
                pp[j].y[dst[i]] = ((ww.x * bj[a] + ww.y * bj[b]) + ww.z * bj[c]) + ww.w * bj[d];
            }
        }
#endif
    }
    wave_lds_sync();      // the next pass overwrites the boxes
}

#ifndef PR_OCC
#define PR_OCC 4
#endif
template <int KIND>
__global__ __launch_bounds__(64 * PR_WAVES, PR_OCC) void polar_from_raw_kernel(PolarRawArgs p) {
    extern __shared__ float lds_all[];
    // the wave index is the same in every lane: say so, or everything derived from it (tile, plane, base address, the
    // normalisation constants) lives in vector registers and is fetched by vector loads
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * PR_WAVES + wave;
    const int g = xcd + 8 * (slot / p.n_tile);
    if (g >= p.n_group) return;
    const PolarTile tl = p.tile[slot % p.n_tile];
    const int bx0 = tl.bx0, by0 = tl.by0, bw = tl.bw, bh = tl.bh;
    const int n_out = tl.rh * tl.aw;
    float* bp = lds_all + wave * (PR_NPL * p.box_stride + 6 * PR_MAXE);      // this wave's boxes ...
    float* tab = bp + PR_NPL * p.box_stride;                                 // ... and its separable-term tables
    unsigned* cx0 = reinterpret_cast<unsigned*>(tab);
    unsigned* cx1 = cx0 + PR_MAXE;
    float* clx1 = tab + 2 * PR_MAXE;
    unsigned* ry0 = reinterpret_cast<unsigned*>(tab + 3 * PR_MAXE);
    unsigned* ry1 = ry0 + PR_MAXE;
    float* rly1 = tab + 5 * PR_MAXE;
    unsigned o01[PR_OUT], o23[PR_OUT];      // the four box offsets of an output, 16 bits each
    int dst[PR_OUT];
    float4 wt[PR_OUT];
#pragma unroll
    for (int i = 0; i < PR_OUT; ++i) {
        const int j = lane + 64 * i;
        dst[i] = -1;
        o01[i] = o23[i] = 0;
        wt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < n_out) {
            const int yy = j / tl.aw, xx = j - yy * tl.aw;
            const int pix = (tl.r0 + yy) * p.Wo + tl.c0 + xx;
            const int4 t = p.taps[pix];       // offsets inside the tile's box (built with the tile table)
            wt[i] = p.wts[pix];
            dst[i] = pix;
            o01[i] = (unsigned)t.x | ((unsigned)t.y << 16);
            o23[i] = (unsigned)t.z | ((unsigned)t.w << 16);
        }
    }
    const int n_plane = p.B * p.C;
    const int pl0 = g * p.planes_per_wave, pl1 = min(n_plane, pl0 + p.planes_per_wave);
    const int n_box = bw * bh;
    const float inv_bw = 1.0f / (float)bw;      // q / bw through the reciprocal: exact for q < 2^12, bw <= 64
    int Hi_prev = -1, Wi_prev = -1, cs_prev = -1;
    int pl = pl0;
    while (pl < pl1) {
        PolarPlane pp[PR_NPL];
        int n = 0, Hi = 0, Wi = 0, cs = 1;
        bool run = true;
#pragma unroll
        for (int j = 0; j < PR_NPL; ++j) {       // the run of planes with the first one's source size (constant j: registers)
            const int plj = min(pl + j, pl1 - 1);
            const int b = plj / p.C, c = plj - b * p.C;
            int h = p.Hi, w = p.Wi, s = 1;
            unsigned long long addr;
            if (p.desc != nullptr) {
                const ImgDesc d = p.desc[b];
                h = (int)d.H; w = (int)d.W; s = (int)d.cs;
                addr = d.ptr + (KIND == 0 ? (unsigned long long)c * h * w * 4 : (unsigned long long)c);
            } else {
                addr = (unsigned long long)p.src + ((unsigned long long)b * p.C + c) * h * w * 4;
            }
            if (j == 0) { Hi = h; Wi = w; cs = s; }
            run = run && pl + j < pl1 && h == Hi && w == Wi && s == cs;
            if (run) n = j + 1;
            // the same in every lane: scalar registers (loads take the form base (SGPR pair) + 32-bit byte offset (VGPR))
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
            pp[j].base = (gmem_u8)(((unsigned long long)hi << 32) | lo);
            pp[j].y = p.y + (size_t)plj * p.Ho * p.Wo;
            pp[j].mean = p.na.mean[c]; pp[j].stdv = p.na.stdv[c]; pp[j].rstd = p.na.rstd[c];
            pp[j].div255 = c < p.na.n_div255;
        }
        // exactly 2:1 in both directions (and, for the 8-byte loads of fp32 sources, an even row length): the table-free path
        const bool half = Hi == 2 * p.size && Wi == 2 * p.size;
        const unsigned px_bytes = KIND == 0 ? 4u : (unsigned)cs, row_bytes = (unsigned)Wi * px_bytes;
        if (!half && (Hi != Hi_prev || Wi != Wi_prev || cs != cs_prev)) {       // wave-uniform
            Hi_prev = Hi; Wi_prev = Wi; cs_prev = cs;
            const float sh = (float)Hi / (float)p.size, sw = (float)Wi / (float)p.size;
            const unsigned es = KIND == 0 ? 4u : (unsigned)cs;      // BYTES between neighbouring pixels of a row
            wave_lds_sync();
            if (lane < bw) {
                float fx = __fmaf_rn(sw, (bx0 + lane) + 0.5f, -0.5f);
                if (fx < 0.f) fx = 0.f;
                const int x0 = (int)fx;
                const int x1 = x0 + ((x0 < Wi - 1) ? 1 : 0);
                cx0[lane] = (unsigned)x0 * es; cx1[lane] = (unsigned)x1 * es; clx1[lane] = fx - x0;
            }
            if (lane < bh) {
                float fy = __fmaf_rn(sh, (by0 + lane) + 0.5f, -0.5f);
                if (fy < 0.f) fy = 0.f;
                const int y0 = (int)fy;
                const int y1 = y0 + ((y0 < Hi - 1) ? 1 : 0);
                ry0[lane] = (unsigned)y0 * (unsigned)Wi * es; ry1[lane] = (unsigned)y1 * (unsigned)Wi * es; rly1[lane] = fy - y0;
            }
            wave_lds_sync();
        }
        const bool norm = p.na.enabled != 0;
#define PR_ARGS pp, norm, lane, bp, p.box_stride, cx0, cx1, clx1, ry0, ry1, rly1, bw, n_box, inv_bw, o01, o23, dst, wt, bx0, by0, row_bytes, px_bytes
        if (half) {
            if (PR_NPL >= 3 && n == 3) polar_planes<KIND, PR_NPL >= 3 ? 3 : 1, true>(PR_ARGS);
            else if (PR_NPL >= 2 && n == 2) polar_planes<KIND, PR_NPL >= 2 ? 2 : 1, true>(PR_ARGS);
            else { n = 1; polar_planes<KIND, 1, true>(PR_ARGS); }
        } else {
            if (PR_NPL >= 3 && n == 3) polar_planes<KIND, PR_NPL >= 3 ? 3 : 1, false>(PR_ARGS);
            else if (PR_NPL >= 2 && n == 2) polar_planes<KIND, PR_NPL >= 2 ? 2 : 1, false>(PR_ARGS);
            else { n = 1; polar_planes<KIND, 1, false>(PR_ARGS); }
        }
#undef PR_ARGS
        pl += n;
    }
}

__global__ void normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int C, size_t hw, size_t total,
                                 NormArgs na) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (idx / hw) % C;
    float v = x[idx];
    if (c < na.n_div255) v = v / 255.f;
    y[idx] = (v - na.mean[c]) / na.stdv[c];
}

// taps: int4 {ia, ib, ic, id} flat offsets into one size x size plane; wts: float4 {wa, wb, wc, wd}.
// One thread per (b, output pixel); the 32-byte LUT entry is read once and reused for all channels.
__global__ void polar_kernel(const float* __restrict__ x, const int4* __restrict__ taps, const float4* __restrict__ wts,
                             float* __restrict__ y, int B, int C, int plane_in, int plane_out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * plane_out) return;
    const int pix = idx % plane_out;
    const int b = (int)(idx / plane_out);
    const int4 t = taps[pix];
    const float4 w = wts[pix];
    for (int c = 0; c < C; ++c) {
        const float* p = x + ((size_t)b * C + c) * plane_in;
        const float v = ((w.x * p[t.x] + w.y * p[t.y]) + w.z * p[t.z]) + w.w * p[t.w];
        y[((size_t)b * C + c) * plane_out + pix] = v;
    }
}

// theta: per image the 2x3 inverse rotation matrix ALREADY divided by (0.5*W, 0.5*H) as torchvision's
// _gen_affine_grid does, laid out [b][k][j] (k = x,y,1 row; j = output coordinate): gx = x*t00 + y*t10 + t20.
// Base grid = half-integer pixel centres (linspace(-W/2+.5, W/2-.5, W): step exactly 1); un-normalisation
// and rounding follow ATen's grid_sampler (align_corners=False, nearbyint, zero outside).
__global__ void rotate_nearest_kernel(const float* __restrict__ x, const float* __restrict__ theta, float* __restrict__ y,
                                      int B, int C, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * H * W) return;
    const int ox = idx % W;
    size_t t = idx / W;
    const int oy = t % H;
    const int b = (int)(t / H);
    const float* th = theta + (size_t)b * 6;
    const float xg = (float)ox + (0.5f - 0.5f * (float)W), yg = (float)oy + (0.5f - 0.5f * (float)H);
    const float gx = __fmaf_rn(yg, th[2], xg * th[0]) + th[4];
    const float gy = __fmaf_rn(yg, th[3], xg * th[1]) + th[5];
    const float fx = ((gx + 1.f) * (float)W - 1.f) / 2.f, fy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float rx = nearbyintf(fx), ry = nearbyintf(fy);
    const bool ok = rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1);
    const size_t src = ok ? (size_t)(int)ry * W + (int)rx : 0;
    for (int c = 0; c < C; ++c) {
        const size_t plane = ((size_t)b * C + c) * H * W;
        y[plane + (size_t)oy * W + ox] = ok ? x[plane + src] : 0.f;
    }
}

int fill_norm(NormArgs& na, int C, const float* mean, const float* stdv, int n_div255) {
    na.enabled = (mean != nullptr && stdv != nullptr);
    na.n_div255 = n_div255;
    for (int c = 0; c < 8; ++c) {
        na.mean[c] = (na.enabled && c < C) ? mean[c] : 0.f;
        na.stdv[c] = (na.enabled && c < C) ? stdv[c] : 1.f;
        // reciprocal for div_exact: positive normal divisor of moderate size whose significand is not all ones
        const float b = na.stdv[c];
        unsigned bits;
        memcpy(&bits, &b, 4);
        const bool ok = b >= 0x1p-10f && b <= 0x1p10f && (bits & 0x7fffffu) != 0x7fffffu;
        na.rstd[c] = ok ? (float)(1.0 / (double)b) : 0.f;
    }
    return 0;
}

}  // namespace

extern "C" {

// mean/stdv are HOST arrays of C floats (nullptr: resize only).
int witw_resize_bilinear_normalize(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, const float* mean,
                                   const float* stdv, int n_div255, void* stream) {
    WITW_CHECK_ARG(x && y, "resize: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0,
                   "resize: bad shape B=%d C=%d %dx%d -> %dx%d", B, C, Hi, Wi, Ho, Wo);
    NormArgs na;
    fill_norm(na, C, mean, stdv, n_div255);
    const size_t total = (size_t)B * Ho * Wo;
    const dim3 grid((unsigned)((total + 255) / 256));
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    if (C == 3)
        hipLaunchKernelGGL(resize_bilinear_norm_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, y, B, C, Hi, Wi, Ho, Wo, sh, sw, na);
    else if (C == 5)
        hipLaunchKernelGGL(resize_bilinear_norm_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, x, y, B, C, Hi, Wi, Ho, Wo, sh, sw, na);
    else
        hipLaunchKernelGGL(resize_bilinear_norm_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, y, B, C, Hi, Wi, Ho, Wo, sh, sw, na);
    WITW_CHECK_LAUNCH("resize_bilinear_normalize");
    return WITW_OK;
}

// Resize (+ normalise) B images of individual sizes in ONE launch. desc: DEVICE array [B][5] of 64-bit words {device address of
// the image, H, W, start column, channels stored per pixel}; kind 0 = float32 planar CHW sources, 1 = uint8 interleaved HWC.
// Output [B,C,Ho,Wo]: the image resized to Ho x Wfull, of which the Wo columns from `start` on (wrapping) are kept
// (Wfull == Wo, start == 0: a plain resize). mean / stdv as in witw_resize_bilinear_normalize.
int witw_resize_bilinear_normalize_batched(const void* desc, float* y, int B, int C, int Ho, int Wo, int Wfull, int kind,
                                           const float* mean, const float* stdv, int n_div255, void* stream) {
    WITW_CHECK_ARG(desc && y, "resize_batched: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && Ho > 0 && Wo > 0 && Wfull >= Wo, "resize_batched: bad shape B=%d C=%d -> %dx%d of %d",
                   B, C, Ho, Wo, Wfull);
    WITW_CHECK_ARG(kind == 0 || kind == 1, "resize_batched: source kind %d unknown (0 float32 CHW, 1 uint8 HWC)", kind);
    NormArgs na;
    fill_norm(na, C, mean, stdv, n_div255);
    const size_t total = (size_t)B * Ho * Wo;
    const dim3 grid((unsigned)((total + 255) / 256));
#define WITW_RB(K, CT) hipLaunchKernelGGL((resize_batched_kernel<K, CT>), grid, dim3(256), 0, (hipStream_t)stream, (const ImgDesc*)desc, y, B, C, Ho, Wo, Wfull, na)
    if (kind == 0) {
        if (C == 3) WITW_RB(0, 3); else if (C == 5) WITW_RB(0, 5); else WITW_RB(0, 0);
    } else {
        if (C == 3) WITW_RB(1, 3); else if (C == 5) WITW_RB(1, 5); else WITW_RB(1, 0);
    }
#undef WITW_RB
    WITW_CHECK_LAUNCH("resize_bilinear_normalize_batched");
    return WITW_OK;
}

int witw_normalize(const float* x, float* y, int B, int C, int H, int W, const float* mean, const float* stdv, int n_div255,
                   void* stream) {
    WITW_CHECK_ARG(x && y && mean && stdv, "normalize: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "normalize: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    NormArgs na;
    fill_norm(na, C, mean, stdv, n_div255);
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, C,
                       (size_t)H * W, total, na);
    WITW_CHECK_LAUNCH("normalize");
    return WITW_OK;
}

// taps/wts: DEVICE LUTs of Ho*Wo entries built on the host in fp64 exactly as the reference does.
int witw_polar_transform(const float* x, const int* taps, const float* wts, float* y, int B, int C, int size, int Ho, int Wo,
                         void* stream) {
    WITW_CHECK_ARG(x && taps && wts && y, "polar_transform: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && size > 0 && Ho > 0 && Wo > 0, "polar_transform: bad shape");
    const size_t total = (size_t)B * Ho * Wo;
    hipLaunchKernelGGL(polar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (const int4*)taps, (const float4*)wts, y, B, C, size * size, Ho * Wo);
    WITW_CHECK_LAUNCH("polar_transform");
    return WITW_OK;
}

// Overhead side of Compose[Resize, ImageNormalization, PolarTransform] (model/cvig_fov.py:393-397) in one launch: raw image ->
// bilinear size x size value -> normalise -> 4-tap polar gather -> y [B,C,Ho,Wo]; bit-identical to
// witw_resize_bilinear_normalize(_batched) + witw_polar_transform. Sources: desc == NULL: src = fp32 [B,C,Hi,Wi]; else desc = the
// DEVICE table of witw_resize_bilinear_normalize_batched (kind 0 fp32 CHW, 1 uint8 HWC; start column unused). taps / wts: the
// DEVICE sampling table of witw_polar_transform, the taps given as offsets (row * box width + column) inside their tile's
// box; tiles: DEVICE int32 [n_tile][8] = {box x0, y0, width, height (bounding box in
// the size x size plane of all taps of the tile), first output row, first output column, rows, columns}; the tiles partition the
// Ho x Wo outputs, rows*columns <= 256, box width and height <= 64, max_box = the largest box width*height (the caller builds
// the table with the sampling table), else -1.
int witw_polar_from_raw(const void* src, const void* desc, int kind, float* y, int B, int C, int Hi, int Wi, int size, int Ho,
                        int Wo, const int* taps, const float* wts, const int* tiles, int n_tile, int max_box, const float* mean,
                        const float* stdv, int n_div255, void* stream) {
    WITW_CHECK_ARG((src || desc) && y && taps && wts && tiles, "polar_from_raw: null pointer");
    WITW_CHECK_ARG(kind == 0 || (kind == 1 && desc), "polar_from_raw: source kind %d (0 float32 CHW, 1 uint8 HWC through a descriptor table)", kind);
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && size > 0 && Ho > 0 && Wo > 0 && (desc || (Hi > 0 && Wi > 0)), "polar_from_raw: bad shape");
    WITW_CHECK_ARG(n_tile > 0 && max_box > 0 && max_box <= 4096, "polar_from_raw: %d tiles, largest box %d pixels: does not fit the LDS",
                   n_tile, max_box);
    PolarRawArgs p;
    p.src = src; p.desc = (const ImgDesc*)desc; p.y = y; p.taps = (const int4*)taps; p.wts = (const float4*)wts;
    p.tile = (const PolarTile*)tiles;
    p.B = B; p.C = C; p.Hi = Hi; p.Wi = Wi; p.size = size; p.Ho = Ho; p.Wo = Wo; p.n_tile = n_tile; p.box_stride = max_box;
    const long long n_plane = (long long)B * C;
    // about 32 waves per CU; a wave keeps its tile's table entries in registers over its run of planes
    // ~24 k (tile, run of planes) units, six rounds of the 4096 resident waves: the boxes of the tiles differ by 4x in area, and short
    // runs let the dispatcher even that out (128 x 3 x 512^2, one box: runs of 24 / 12 / 8 / 4 / 3 planes: 152 / 139 / 131 / 128 / 131 us)
    p.planes_per_wave = (int)((n_plane * n_tile + 24575) / 24576);
    if (const char* e = getenv("WITW_PR_PPW")) p.planes_per_wave = atoi(e);      // diagnostic: planes a wave takes per tile
    if (p.planes_per_wave < 1) p.planes_per_wave = 1;
    if (n_plane >= PR_NPL) p.planes_per_wave = (p.planes_per_wave + PR_NPL - 1) / PR_NPL * PR_NPL;      // whole passes of PR_NPL planes
    p.n_group = (int)((n_plane + p.planes_per_wave - 1) / p.planes_per_wave);
    fill_norm(p.na, C, mean, stdv, n_div255);
    const long long slots = (long long)((p.n_group + 7) / 8) * n_tile;       // (run of planes, tile) pairs per XCD
    const unsigned grid = 8u * (unsigned)((slots + PR_WAVES - 1) / PR_WAVES);
    const size_t lds = (size_t)PR_WAVES * (PR_NPL * max_box + 6 * PR_MAXE) * 4;
    if (kind == 0)
        hipLaunchKernelGGL(polar_from_raw_kernel<0>, dim3(grid), dim3(64 * PR_WAVES), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(polar_from_raw_kernel<1>, dim3(grid), dim3(64 * PR_WAVES), lds, (hipStream_t)stream, p);
    WITW_CHECK_LAUNCH("polar_from_raw");
    return WITW_OK;
}

// The same 4-tap gather for any sampling table (bilinear_interpolate of the reference, model/cvig_fov.py:156-183, on
// arbitrary coordinates): taps are flat offsets into a plane of plane_in elements, n_out output samples per plane.
int witw_bilinear_gather(const float* x, const int* taps, const float* wts, float* y, int B, int C, long long plane_in,
                         long long n_out, void* stream) {
    WITW_CHECK_ARG(x && taps && wts && y, "bilinear_gather: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && plane_in > 0 && plane_in <= 0x7fffffffLL && n_out > 0 && n_out <= 0x7fffffffLL,
                   "bilinear_gather: bad shape");
    const size_t total = (size_t)B * n_out;
    hipLaunchKernelGGL(polar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (const int4*)taps, (const float4*)wts, y, B, C, (int)plane_in, (int)n_out);
    WITW_CHECK_LAUNCH("bilinear_gather");
    return WITW_OK;
}

// x, y: [B,C,H,W] fp32 (y != x); theta: DEVICE [B,2,3]^T rescaled matrices, see rotate_nearest_kernel.
int witw_rotate_nearest(const float* x, const float* theta, float* y, int B, int C, int H, int W, void* stream) {
    WITW_CHECK_ARG(x && theta && y && x != y, "rotate_nearest: null or aliased pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "rotate_nearest: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(rotate_nearest_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       theta, y, B, C, H, W);
    WITW_CHECK_LAUNCH("rotate_nearest");
    return WITW_OK;
}

}  // extern "C"
