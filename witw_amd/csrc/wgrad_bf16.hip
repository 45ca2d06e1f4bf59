// Weight gradient of the 3x3 convolution on v_mfma_f32_32x32x16_bf16 (gfx950): the mixed-precision training
// step of the FOV_DSM encoder (bf16 operands, fp32 accumulate, fp32 gradients out).
//
// Reference semantics: autograd through torch.nn.Conv2d in the training loop of model/cvig_fov.py:447-460
// (trainable layers :275-278; model/cvig_semantic.py:301-309 adds layer 0), i.e.
//   dW[co][ci][kh][kw] = sum_{b,h,w} dZ[b,h,w,co] * Xpad[b, h*SH+kh-1, w+kw-1, ci]
// with dZ the gradient at the conv output. The contraction runs over (image, pixel); a bf16 MFMA lane needs
// EIGHT consecutive k values of its row/column, which in NHWC (channels innermost) are 8 far-apart pixels.
// Instead of transposing through LDS, both operands are re-laid out once per layer (an HBM-bound pass over
// tensors of at most a few 100 MB) into the BATCH-OCTET layout
//   [B/8][H][W][C][8 images]   (16 bytes per (pixel, channel) slot)
// so that the 8 k values of a lane are the 8 images of one octet at ONE pixel: a filter tap is then a pure
// pixel offset of the X operand, every operand fragment is one conflict-free ds_read_b128, and the two lane
// halves of the MFMA (k = 0-7 / 8-15) take two horizontally adjacent pixels.
//
// Workgroup = 8 waves (two per SIMD), tile 64 (ci) x 128 (co) of ALL 9 taps: wave (wm, wn) owns 32 ci x 32 co = 9
// accumulator tiles (144 registers; 18 tiles per wave do not fit the 256 accumulator registers). One K chunk = one image octet x R output rows x 8 output columns: the (R-1)*SH+3 halo rows x 10 columns
// of X and the R x 8 pixels of dZ move into LDS by LDS-DMA (buffer_load ... lds: a (pixel, 64 channels) run is
// 1 KB contiguous in both HBM and LDS; padding = out-of-range offsets, which store zeros), double buffered.
// Split-K partials go to a workspace and are summed in a fixed order (bitwise reproducible).
// The bias gradient db[co] = sum dZ rides along: the waves of the first ci tile that own ci rows 0-31 feed the dZ
// fragment they hold anyway to two v_mfma_f32_16x16x32_bf16 against a constant 'ones in row 0' operand (+1/9 MFMA
// time on 1/(Cin/64) of the workgroups) instead of a separate pass over dZ.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int WB_P = 8;                 // output columns per chunk
constexpr int WB_XC = WB_P + 2;         // halo columns
constexpr int WB_TM = 64, WB_TN = 128;  // ci x co tile of a workgroup
constexpr unsigned OOR = 0x80000000u;

__device__ __forceinline__ i32x4 raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// one wave instruction of LDS-DMA: lane l moves 16 B from rs[voff_l + soff] to LDS byte lds_addr + 16*l
// (out-of-range lanes store zeros); see conv3x3_bf16.hip for why this is inline assembly
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");
#endif
}

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

struct WgradBfArgs {
    const unsigned short* x;    // [B8][H][W][Cin][8]   bf16, batch-octet layout
    const unsigned short* dz;   // [B8][Ho][Wo][Cout][8]
    float* ws;                  // [splits][9][Cin][Cout]
    float* bias_part;           // nullptr, or [splits][Cout] partial bias gradients
    int B8, H, W, Cin, Cout, Ho, Wo;
    int circ;
    int nseg;                   // column segments of WB_P per output row
    int nrg;                    // row groups of R per image octet
    int chunks;                 // B8 * nrg * nseg
    int cps;                    // chunks per split
};

template <int SH, int R>
__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16_kernel(WgradBfArgs p) {
    constexpr int NW = 8;
    constexpr int XR = (R - 1) * SH + 3;              // halo rows
    constexpr int NXI = XR * WB_XC;                   // X DMA instructions per stage (one per halo pixel: 64 ci slots)
    constexpr int NZI = R * WB_P * 2;                 // dZ DMA instructions per stage (one per pixel and co half)
    constexpr int X_S = NXI * 64;                     // 16-B slots
    constexpr int Z_S = NZI * 64;
    constexpr int STAGE_S = X_S + Z_S;
    static_assert(2 * STAGE_S * 16 <= 160 * 1024, "two stages must fit the LDS");
    __shared__ u32x4 stageA[STAGE_S];
    __shared__ u32x4 stageB[STAGE_S];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6) & (NW - 1);
    const int l31 = lane & 31, kg = lane >> 5;
    const int wm = wave_u & 1, wn = wave_u >> 1;
    const int ci0 = blockIdx.x * WB_TM, co0 = blockIdx.y * WB_TN, split = blockIdx.z;
    const int c_begin = split * p.cps;
    const int c_end = min(p.chunks, c_begin + p.cps);

    const i32x4 x_rs = raw_rsrc(p.x, (unsigned)((size_t)p.B8 * p.H * p.W * p.Cin * 16u));
    const i32x4 z_rs = raw_rsrc(p.dz, (unsigned)((size_t)p.B8 * p.Ho * p.Wo * p.Cout * 16u));
    const unsigned x_lane = (ci0 + lane < p.Cin) ? (unsigned)lane * 16u : OOR;
    unsigned z_lane[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) z_lane[h] = (co0 + h * 64 + lane < p.Cout) ? (unsigned)lane * 16u : OOR;

    // chunk c -> stage s: this wave's share of the DMA instructions
    auto stage = [&](int c, u32x4* s) {
        const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(s));
        const int seg = c % p.nseg;
        const int t = c / p.nseg;
        const int rg = t % p.nrg, b8 = t / p.nrg;
        const int h0 = rg * R, w0 = seg * WB_P;
#pragma unroll
        for (int i = 0; i < (NXI + NW - 1) / NW; ++i) {
            const int j = wave_u + NW * i;
            if (NXI % NW == 0 || j < NXI) {
                const int r = j / WB_XC, cc = j - r * WB_XC;
                const int gr = h0 * SH - 1 + r;
                int gc = w0 - 1 + cc;
                bool ok = gr >= 0 && gr < p.H;
                if (p.circ) {               // only columns -1 and W wrap; columns past W pair with zero dZ pixels
                    if (gc < 0) gc += p.W;
                    else if (gc >= p.W) gc -= p.W;
                }
                ok = ok && gc >= 0 && gc < p.W;
                const unsigned soff = ok ? (unsigned)(((((size_t)b8 * p.H + gr) * p.W + gc) * p.Cin + ci0) * 16u) : 0u;
                dma16(x_rs, lds + (unsigned)j * 1024u, ok ? x_lane : OOR, soff);
            }
        }
        static_assert(NZI % NW == 0, "dZ instructions split evenly over the waves");
#pragma unroll
        for (int i = 0; i < NZI / NW; ++i) {
            const int j = wave_u + NW * i;
            const int half = j & 1, pp = j >> 1;
            const int rr = pp / WB_P, px = pp - rr * WB_P;
            const int h = h0 + rr, w = w0 + px;
            const bool ok = h < p.Ho && w < p.Wo;
            const unsigned soff = ok ? (unsigned)(((((size_t)b8 * p.Ho + h) * p.Wo + w) * p.Cout + co0 + half * 64) * 16u) : 0u;
            dma16(z_rs, lds + (unsigned)(X_S + j * 64) * 16u, ok ? z_lane[half] : OOR, soff);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int abase = kg * 64 + wm * 32 + l31;               // X slot of (tile pixel 0 + kg, this lane's ci)
    const int bbase = X_S + kg * 128 + wn * 32 + l31;        // dZ slot of (pixel 0 + kg, this lane's co)

    // all MFMAs of one staged chunk: R rows x 4 pixel pairs (k-steps) x 9 taps; the fragments of k-step i+1 are
    // read while the 9 MFMAs of k-step i run
    constexpr int KS = R * WB_P / 2;
    static_assert(KS % 2 == 0, "k-steps are processed in pairs");
    u32x4 fa[2][9], fb[2];
    // bias gradient: the dZ fragment (lane l: co = l & 31, pixel = l >> 5, 8 images) read as the B operand of the
    // 16x16x32 shape has k groups (l >> 4) = {co 0-15 px0, co 16-31 px0, co 0-15 px1, co 16-31 px1}; ones in row 0 of
    // A on groups {0,2} / {1,3} sum the images and both pixels of co 0-15 / 16-31 into row 0 of the result
    const bool do_bias = p.bias_part != nullptr && blockIdx.x == 0 && wm == 0;     // wave-uniform
    f32x4 accb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    u32x4 ones[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned v = ((lane & 15) == 0 && ((lane >> 4) & 1) == h) ? 0x3F803F80u : 0u;
        ones[h] = (u32x4){v, v, v, v};
    }
    auto read_frags = [&](int set, const u32x4* s, int ks) {
        const int rr = ks / (WB_P / 2), kp = ks % (WB_P / 2);
        const u32x4* ap = s + abase + (rr * SH * WB_XC + 2 * kp) * 64;
        fb[set] = s[bbase + (rr * WB_P + 2 * kp) * 128];
#pragma unroll
        for (int t = 0; t < 9; ++t) fa[set][t] = ap[((t / 3) * WB_XC + (t % 3)) * 64];
    };
    auto mfma_step = [&](int set) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][t]),
                                                             __builtin_bit_cast(bf16x8, fb[set]), acc[t], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                accb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones[h]),
                                                                  __builtin_bit_cast(bf16x8, fb[set]), accb[h], 0, 0, 0);
        }
    };
    auto compute = [&](const u32x4* s) {
        read_frags(0, s, 0);
#pragma unroll 1
        for (int ks = 0; ks < KS; ks += 2) {
            read_frags(1, s, ks + 1);
            mfma_step(0);
            if (ks + 2 < KS) read_frags(0, s, ks + 2);
            mfma_step(1);
        }
    };

    if (c_begin < c_end) {
        stage(c_begin, stageA);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMA has landed
        __syncthreads();
        for (int c = c_begin; c < c_end; c += 2) {
            if (c + 1 < c_end) stage(c + 1, stageB);
            compute(stageA);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            if (c + 1 < c_end) {
                if (c + 2 < c_end) stage(c + 2, stageA);
                compute(stageB);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
            }
        }
    }

    // ---- partial tile -> workspace [split][tap][ci][co]
    float* out = p.ws + (size_t)split * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int co = co0 + wn * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (ci < p.Cin && co < p.Cout) out[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][r];
        }
    }
    if (do_bias && lane < 16) {          // row 0 of the 16x16 results: lanes 0-15, register 0
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int co = co0 + wn * 32 + h * 16 + lane;
            if (co < p.Cout) p.bias_part[(size_t)split * p.Cout + co] = accb[h][0];
        }
    }
}

// dW[co][ci][kh][kw] (+)= sum_split ws[split][tap][ci][co]; one thread per (tap, ci, co), co fastest.
// Threads past the weight elements sum the bias partials: db[co] (+)= sum_split bias_part[split][co].
__global__ void wgrad_bf16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cin, int Cout, int splits,
                                         int accumulate, int cin_real, const float* __restrict__ bias_part,
                                         float* __restrict__ db) {
    const size_t n = (size_t)9 * Cin * Cout;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) {
        const size_t co = idx - n;
        if (db != nullptr && co < (size_t)Cout) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += bias_part[(size_t)k * Cout + co];
            db[co] = accumulate ? db[co] + s : s;
        }
        return;
    }
    const int co = idx % Cout;
    const size_t t = idx / Cout;
    const int ci = t % Cin;
    const int tap = (int)(t / Cin);
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * n + idx];
    if (ci >= cin_real) return;
    float* d = dw + ((size_t)co * cin_real + ci) * 9 + tap;
    *d = accumulate ? (*d + s) : s;
}

// NHWC bf16 [B][HW][C] -> batch-octet [ceil(B/8)][HW][C][8] (images past B are zeros). One thread per
// (octet, pixel, channel octet): 8 loads of 16 B (8 channels of one image), an 8x8 transpose in registers,
// 8 stores of 16 B (one channel, 8 images) = 128 contiguous bytes.
__global__ void nhwc_to_octet_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int B, size_t HW, int C,
                                     size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int C8 = C >> 3;
    const int c8 = idx % C8;
    const size_t t = idx / C8;
    const size_t pix = t % HW;
    const size_t b8 = t / HW;
    u16x8 in[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const size_t b = b8 * 8 + i;
        if (b < (size_t)B)
            in[i] = *reinterpret_cast<const u16x8*>(x + ((b * HW + pix) * C + (size_t)c8 * 8));
        else
            in[i] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    u16x8* out = reinterpret_cast<u16x8*>(y + (((b8 * HW + pix) * C + (size_t)c8 * 8) * 8));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        u16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = in[i][j];
        out[j] = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the same weight gradient straight from the NHWC tensors the forward / dgrad kernels write -- no batch-octet copies
// (nhwc_to_octet_kernel was 1.05 ms of the 15.0 ms bf16 training step, 2.5 ms of cvig_semantic's 29 ms; profiles/r05_*).
//
// The contraction index k of the MFMA is now the PIXEL: one k-step = 16 consecutive output columns of one row of one image
// (lanes 0-31 take columns 0-7, lanes 32-63 columns 8-15). In NHWC a lane's 8 k values are 8 pixels = 8 far-apart 2-byte
// elements; gfx950's ds_read_b64_tr_b16 does that transpose on the way out of the LDS: a group of 16 lanes reads a block of
// 4 rows (pixels) x 16 columns (channels) and each lane receives one channel of the 4 pixels. Two such reads are one MFMA
// operand. The LDS images are plain [pixel][channel] rows (X: 64 ci = 128 B per pixel, pitch XP = 24 pixels per halo row;
// dZ: 128 co = 256 B per pixel), filled by LDS-DMA in 16-byte chunks. A DMA lane may fetch ANY global chunk, so the bank
// swizzle costs nothing: LDS chunk position c of pixel p holds channel chunk c ^ (bit 1 of p << 2) (X) / c ^ ((p & 3) << 2)
// (dZ), which makes every transposed read conflict-free (4 consecutive pixels x 64 B land on 64 distinct banks).
// A filter tap is a pixel offset of the X read, as before. One stage = one image x R output rows x 16 columns: its
// ((R-1)*SH+3) x 18 halo pixels of X and R x 16 pixels of dZ; zero padding / ragged edges = out-of-range DMA lanes (zeros).
// Workgroup tile, wave roles, split-K workspace and the fixed-order reduction are those of the octet kernel above. The
// workgroup -> (tile, split) map is XCD-aware: all tiles of one split share an XCD (blocks b and b + 8 do), so the X and dZ
// slices a split streams are fetched from HBM once and re-used out of that XCD's L2 by the other ci / co tiles (the (x, y, z)
// grid of the octet kernel put one ci tile per XCD: every XCD streamed all of dZ).
struct WgradNhArgs {
    const unsigned short* x;    // [B][H][W][Cin]    bf16 NHWC
    const unsigned short* dz;   // [B][Ho][Wo][Cout] bf16 NHWC
    float* ws;                  // [splits][9][Cin][Cout]
    float* bias_part;           // nullptr, or [splits][Cout]
    int B, H, W, Cin, Cout, Ho, Wo;
    int circ;
    int nseg, nrg, chunks, cps;
    int tiles_ci, tiles_co, splits;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int NH_P = 16;                // output columns per k-step
constexpr int NH_XP = 24;               // pixel pitch of a halo row in the LDS (18 used: 3 DMA instructions of 8 pixels per row)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {      // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt left alone); N <= 63
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

// Wave roles: NWM x NWN waves cover the 32-channel slabs of the tile (ci x co) that EXIST -- a layer with Cin <= 32 has one ci slab
// (NWM = 1), one with Cout <= 64 / <= 32 two / one co slabs -- and the remaining factor KS = 8 / (NWM * NWN) splits the stage's R
// rows (k-steps) into KS contiguous blocks, one per wave group: every wave has MFMA work on its own rows instead of multiplying
// zeros (cvig_semantic's layer 0, 5 -> 64 channels on 128 x 512 maps, ran 2 of 8 waves' worth of useful MFMAs and was MFMA-bound
// where it should be HBM-bound). The wave groups' sums are added in the LDS at the end of the kernel (fixed order).
template <int SH, int R, int NS, int NWM, int NWN>
__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16_nhwc_kernel(WgradNhArgs p) {
    constexpr int NW = 8;
    constexpr int KS = NW / (NWM * NWN);              // wave groups along k (rows of the stage)
    constexpr int RW = R / KS;                        // output rows per wave and stage
    static_assert(NWM * NWN * KS == NW && RW * KS == R && RW >= 1, "8 waves = NWM x NWN x KS; the rows split evenly");
    constexpr int XR = (R - 1) * SH + 3;              // halo rows
    constexpr int NXI = XR * 3;                       // X DMA instructions per stage (8 pixels x 128 B each)
    constexpr int NZI = R * 4;                        // dZ DMA instructions per stage (4 pixels x 256 B each)
    constexpr int X_B = NXI * 1024, Z_B = NZI * 1024; // bytes
    constexpr int STAGE_B = X_B + Z_B;
    static_assert(NS >= 2 && NS * STAGE_B <= 160 * 1024, "the stage ring must fit the LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6) & (NW - 1);
    const int l31 = lane & 31, kg = lane >> 5;
    const int wm = wave_u % NWM, wn = (wave_u / NWM) % NWN, ks = wave_u / (NWM * NWN);
    // XCD-aware map: physical block b runs on XCD b % 8 (observed round-robin; speed only). Logical ids are dealt so that each XCD
    // gets a CONTIGUOUS range of them, and logical id = split * tiles + tile: the tiles of a split share an XCD.
    const int tiles = p.tiles_ci * p.tiles_co;
    const int nblk = tiles * p.splits;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int logical = xcd * (nblk >> 3) + min(xcd, nblk & 7) + slot;
    const int split = logical / tiles, tile = logical - split * tiles;
    const int ci0 = (tile % p.tiles_ci) * WB_TM, co0 = (tile / p.tiles_ci) * WB_TN;
    const int c_begin = split * p.cps;
    const int c_end = min(p.chunks, c_begin + p.cps);

    const i32x4 x_rs = raw_rsrc(p.x, (unsigned)((size_t)p.B * p.H * p.W * p.Cin * 2u));
    const i32x4 z_rs = raw_rsrc(p.dz, (unsigned)((size_t)p.B * p.Ho * p.Wo * p.Cout * 2u));
    // DMA lane roles. X instruction: 8 pixels x 8 chunks; lane -> pixel (lane >> 3) of the 8, LDS chunk position lane & 7, which
    // holds channel chunk (lane & 7) ^ (bit 1 of the pixel's column index << 2); the column index of instruction part q is
    // 8 q + (lane >> 3), whose bit 1 is that of (lane >> 3).
    const int xpx = lane >> 3;
    const int xchunk = (lane & 7) ^ (((xpx >> 1) & 1) << 2);
    const bool x_ci_ok = ci0 + xchunk * 8 < p.Cin;
    const unsigned x_lane_b = (unsigned)(ci0 + xchunk * 8) * 2u;
    // dZ instruction: 4 pixels x 16 chunks; lane -> pixel (lane >> 4), LDS chunk position lane & 15 holding chunk ^ ((pixel & 3) << 2)
    const int zpx = lane >> 4;
    const int zchunk = (lane & 15) ^ (zpx << 2);
    const bool z_co_ok = co0 + zchunk * 8 < p.Cout;
    const unsigned z_lane_b = (unsigned)(zpx * p.Cout + co0 + zchunk * 8) * 2u;

    // One stage = NXW + NZW DMA instructions of this wave ("slots"; waves past NXI % NW / NZI % NW have one less). stage_begin fixes
    // the stage's scalars, stage_slot(k) issues slot k: the main loop spreads the slots of the NEXT stage over the first rows of the
    // current stage's MFMAs (round 5, second half: issued as one block in front of the MFMAs, the 8 waves' 62 instructions queue at
    // the texture path and every wave sits at its issue until they are taken -- outside an MFMA's shadow a DMA instruction costs
    // far more than the ~10 cycles it costs inside one, as the spectral match's epilogue showed).
    constexpr int NXW = (NXI + NW - 1) / NW, NZW = (NZI + NW - 1) / NW, NSLOT = NXW + NZW;
    static_assert(NZI % NW == 0 || NZI < NW, "dZ instructions split evenly over the waves");
    unsigned st_base = 0;
    int st_b = 0, st_h0 = 0, st_w0 = 0;
    auto stage_begin = [&](int c, int buf) {
        st_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(lds)) + (unsigned)buf * STAGE_B;
        const int seg = c % p.nseg;
        const int t = c / p.nseg;
        const int rg = t % p.nrg;
        st_b = t / p.nrg;
        st_h0 = rg * R;
        st_w0 = seg * NH_P;
    };
    auto stage_slot = [&](int k) {      // k compile-time after unrolling
        if (k < NXW) {
            const int j = wave_u + NW * k;                       // wave-uniform
            if (NXI % NW == 0 || j < NXI) {
                const int r = j / 3, part = j - r * 3;
                const int gr = st_h0 * SH - 1 + r;
                const int col = part * 8 + xpx;                  // 0..23 (18 used)
                int gc = st_w0 - 1 + col;
                if (p.circ) {                                    // only columns -1 and W wrap; columns past W pair with zero dZ pixels
                    if (gc < 0) gc += p.W;
                    else if (gc == p.W) gc = 0;
                }
                const bool ok = x_ci_ok && col < NH_P + 2 && gc >= 0 && gc < p.W && gr >= 0 && gr < p.H;
                const unsigned soff = (gr >= 0 && gr < p.H) ? (unsigned)(((size_t)st_b * p.H + gr) * p.W * p.Cin * 2u) : 0u;
                dma16(x_rs, st_base + (unsigned)j * 1024u, ok ? (unsigned)(gc * p.Cin) * 2u + x_lane_b : OOR, soff);
            }
        } else {
            const int j = wave_u + NW * (k - NXW);
            if (NZI % NW == 0 || j < NZI) {
                const int r = j >> 2, c4 = (j & 3) * 4;
                const int h = st_h0 + r, w = st_w0 + c4;         // this instruction's first pixel
                const bool ok = z_co_ok && h < p.Ho && w + zpx < p.Wo;
                const unsigned soff = (h < p.Ho && w < p.Wo) ? (unsigned)((((size_t)st_b * p.Ho + h) * p.Wo + w) * p.Cout * 2u) : 0u;
                dma16(z_rs, st_base + (unsigned)(X_B + j * 1024), ok ? z_lane_b : OOR, soff);
            }
        }
    };
    auto stage = [&](int c, int buf) {      // a whole stage at once (the ring's first NS - 1 stages)
        stage_begin(c, buf);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) stage_slot(k);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read lane roles: group G = lane >> 4 (G & 1: channel half of the wave's 32, G >> 1 = kg: pixel half of the k-step),
    // q = (lane >> 2) & 3: the pixel of the 4-row block whose address this lane supplies, pp = lane & 3: which 8 bytes of its 32
    const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    unsigned a_lane[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int col = 8 * kg + q + kw;                                     // + 4 h per read half; bit 1 unaffected by 8 kg + 4 h
        const int chunk = (4 * wm + 2 * (G & 1) + (pp >> 1)) ^ (((col >> 1) & 1) << 2);
        a_lane[kw] = (unsigned)(col * 128 + chunk * 16 + 8 * (pp & 1) + ks * RW * SH * NH_XP * 128);       // this wave group's first halo row
    }
    const unsigned b_lane = (unsigned)(X_B + (8 * kg + q) * 256 + (((4 * wn + 2 * (G & 1) + (pp >> 1)) ^ (q << 2)) * 16) + 8 * (pp & 1) +
                                       ks * RW * NH_P * 256);

    const bool do_bias = p.bias_part != nullptr && ci0 == 0 && wm == 0;     // wave-uniform
    f32x4 accb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    u32x4 ones[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned v = ((lane & 15) == 0 && ((lane >> 4) & 1) == h) ? 0x3F803F80u : 0u;
        ones[h] = (u32x4){v, v, v, v};
    }

    auto tr = [&](const unsigned char* sbase, unsigned off) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sbase + off));
    };
    auto frag = [&](const unsigned char* sbase, unsigned off) -> bf16x8 {       // 8 consecutive pixels of this lane's channel
        const s16x4 lo = tr(sbase, off), hi = tr(sbase, off + 4 * 128);
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto fragz = [&](const unsigned char* sbase, unsigned off) -> bf16x8 {
        const s16x4 lo = tr(sbase, off), hi = tr(sbase, off + 4 * 256);
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // all MFMAs of one staged chunk: R k-steps (one per output row) x 9 taps. The X fragments of halo row j serve every output row
    // whose window holds j (taps kh = j - r*SH): with SH = 1 an output row needs only ONE new halo row = 3 fragments (6 transposed
    // reads) + its dZ fragment (2 reads) for 9 MFMAs, not 9 + 1 fragments. The unrolled loop names every halo row's fragments once
    // (fx[j][kw]); those of the next output row are requested in front of the current row's MFMAs. Rows past Ho were staged as zeros
    // (out-of-range DMA lanes) and are computed like the others.
    constexpr int ROWS_PF = RW >= 4 ? RW / 2 : RW;                     // rows of a stage that carry the next stage's DMA slots ...
    constexpr int PER_ROW = (NSLOT + ROWS_PF - 1) / ROWS_PF;            // ... this many each (with RW >= 4 the last one has RW / 2 rows to land)
    constexpr int EVERY = PER_ROW <= 4 ? 2 : 1;                         // one slot behind every second MFMA of the row, or behind every one
    static_assert(PER_ROW <= 9, "a row's 9 MFMAs carry at most 9 DMA slots");
    // Spread only where every wave owns all R rows of the stage (KS = 1: the 128-channel layers, 2-3 % faster); where the waves split
    // the rows (narrow layers: few MFMAs per wave and stage) the block in front of the MFMAs measured 2-7 % faster and is kept
    constexpr bool SPREAD = KS == 1;
    auto compute = [&](int buf, bool prefetch) {
        const unsigned char* sb = lds + buf * STAGE_B;
        constexpr int XRW = (RW - 1) * SH + 3;      // halo rows of this wave's RW output rows
        bf16x8 fx[XRW][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) fx[j][kw] = frag(sb, a_lane[kw] + (unsigned)(j * NH_XP * 128));
        bf16x8 fb = fragz(sb, b_lane);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            bf16x8 fbn = fb;
            if (r + 1 < RW) {
                fbn = fragz(sb, b_lane + (unsigned)((r + 1) * NH_P * 256));
#pragma unroll
                for (int j = r * SH + 3; j < (r + 1) * SH + 3; ++j)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) fx[j][kw] = frag(sb, a_lane[kw] + (unsigned)(j * NH_XP * 128));
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[r * SH + t / 3][t % 3], fb, acc[t], 0, 0, 0);
                // the next stage's DMA slots of this row, one behind every other MFMA
                if (SPREAD && r < ROWS_PF && (t % EVERY) == EVERY - 1 && r * PER_ROW + t / EVERY < min(NSLOT, (r + 1) * PER_ROW)) {
                    if (prefetch) stage_slot(r * PER_ROW + t / EVERY);
                }
            }
            if (do_bias) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    accb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones[h]), fb, accb[h], 0, 0, 0);
            }
            fb = fbn;
        }
    };

    // Ring of NS stages, one barrier per chunk: chunk c + NS - 1 is requested into the buffer chunk c - 1 has just left, so a DMA
    // has NS - 1 compute phases to land. This wave's own DMA instructions per stage: the waves share NXI and NZI unevenly.
    constexpr int LO = NXI / NW + NZI / NW;                                  // instructions per stage of a wave ...
    static_assert(NXI % NW == 0 || NZI % NW == 0, "at most one of the two instruction counts is uneven");
    const bool extra = wave_u < NXI % NW || wave_u < NZI % NW;               // ... plus one for these waves
    if (c_begin < c_end) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (c_begin + s < c_end) stage(c_begin + s, s);
        int buf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            // chunk c has landed when at most the NS - 2 younger stages' instructions of this wave are outstanding (in-order return)
            if (NS > 2 && c + NS - 2 < c_end) {
                if (extra) wait_vmcnt<(NS - 2) * (LO + 1)>();
                else wait_vmcnt<(NS - 2) * LO>();
            } else {
                wait_vmcnt<0>();
            }
            __syncthreads();                         // ... everyone's has, and everyone is done reading chunk c - 1's buffer
            const bool prefetch = c + NS - 1 < c_end;
            if (prefetch) {
                if (SPREAD) stage_begin(c + NS - 1, buf == 0 ? NS - 1 : buf - 1);
                else stage(c + NS - 1, buf == 0 ? NS - 1 : buf - 1);
            }
            compute(buf, prefetch);
            buf = buf + 1 == NS ? 0 : buf + 1;
        }
    }

    // ---- the KS wave groups' sums of one slab pair are added in the LDS, group 1, 2, ... onto group 0 (a fixed order), one tap at
    // a time (8 waves x 16 registers x 64 lanes x 4 B = 32 KB): the workspace keeps ONE partial per workgroup
    if (KS > 1) {
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int t = 0; t < 10; ++t) {                 // t = 9: the two bias tiles
            __syncthreads();                           // the stage buffers / the previous tap's exchange have been read
            if (ks > 0) {
                if (t < 9) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(wave_u * 16 + r) * 64 + lane] = acc[t][r];
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) red[(wave_u * 16 + h * 4 + r) * 64 + lane] = accb[h][r];
                }
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll 1
                for (int k = 1; k < KS; ++k) {
                    const int src = wave_u + k * (NWM * NWN);
                    if (t < 9) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] += red[(src * 16 + r) * 64 + lane];
                    } else {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int r = 0; r < 4; ++r) accb[h][r] += red[(src * 16 + h * 4 + r) * 64 + lane];
                    }
                }
            }
        }
        if (ks > 0) return;
    }

    // ---- partial tile -> workspace [split][tap][ci][co]
    float* out = p.ws + (size_t)split * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int co = co0 + wn * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (ci < p.Cin && co < p.Cout) out[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][r];
        }
    }
    if (do_bias && lane < 16) {          // row 0 of the 16x16 results: lanes 0-15, register 0
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int co = co0 + wn * 32 + h * 16 + lane;
            if (co < p.Cout) p.bias_part[(size_t)split * p.Cout + co] = accb[h][0];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x32_bf16 (round 5; full-width layers with Wo % 32 == 0: layers 17-23 of FOV_DSM). On random
// bf16 data the chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal FLOPs and operand traffic
// (MI355X_MICROARCH.md, DVFS give-back (7): x1.12-1.15 in bare loops; conv3x3_bf16_s16_kernel gained 6.5 % of the step from it).
//   k-step = 32 consecutive output columns of one row (k-group g = lane >> 4 takes columns 8 g .. 8 g + 7);
//   stage  = one image x R rows x 32 columns: X halo ((R-1)*SH+3) x 34 pixels at a pitch of 40 (5 DMA instructions per row),
//            dZ R x 32 pixels -- the same LDS bytes and DMA instruction counts as the 32x32x16 kernel's 8 x 16 stage;
//   waves  = 4 (16 input channels each) x 2 (64 output channels each): a wave's 16 ci x 64 co x 9 taps are 9 x 4 accumulator
//            tiles of 4 registers (144, as before); per k-step it reads one new halo row (3 fragments = 6 transposed reads) and
//            four dZ fragments (8 reads) for 36 MFMAs;
//   swizzles: a half-wave's transposed read now covers 8 pixels (p, p + 8 for four consecutive p), so bit 3 of the column joins the
//            XOR: X chunk ^ (bit1(col) << 1 | bit3(col) << 2), dZ chunk ^ ((col & 3) << 1 | bit3(col) << 3) -- conflict-free.
// Everything else (split-K map, workspace, reduction, bias through a ones operand) is the 32x32x16 kernel's.
constexpr int N16_P = 32;               // output columns per k-step
constexpr int N16_XP = 40;              // pixel pitch of a halo row (34 used)

template <int SH, int R>
__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16_nhwc16_kernel(WgradNhArgs p) {
    constexpr int NW = 8;
    constexpr int XR = (R - 1) * SH + 3;
    constexpr int NXI = XR * 5;                       // X DMA instructions per stage (8 pixels x 128 B each)
    constexpr int NZI = R * 8;                        // dZ DMA instructions per stage (4 pixels x 256 B each)
    constexpr int X_B = NXI * 1024, Z_B = NZI * 1024;
    constexpr int STAGE_B = X_B + Z_B;
    static_assert(2 * STAGE_B <= 160 * 1024, "two stages must fit the LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6) & (NW - 1);
    const int wm = wave_u & 3, wn = wave_u >> 2;
    const int tiles = p.tiles_ci * p.tiles_co;
    const int nblk = tiles * p.splits;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int logical = xcd * (nblk >> 3) + min(xcd, nblk & 7) + slot;
    const int split = logical / tiles, tile = logical - split * tiles;
    const int ci0 = (tile % p.tiles_ci) * WB_TM, co0 = (tile / p.tiles_ci) * WB_TN;
    const int c_begin = split * p.cps;
    const int c_end = min(p.chunks, c_begin + p.cps);

    const i32x4 x_rs = raw_rsrc(p.x, (unsigned)((size_t)p.B * p.H * p.W * p.Cin * 2u));
    const i32x4 z_rs = raw_rsrc(p.dz, (unsigned)((size_t)p.B * p.Ho * p.Wo * p.Cout * 2u));
    const int xpx = lane >> 3;                        // pixel of an X instruction's 8
    const int xc_l = (lane & 7) ^ (((xpx >> 1) & 1) << 1);      // channel chunk, lane part of the swizzle (bit 3 of the column: per instruction)
    const int zpx = lane >> 4;                        // pixel of a dZ instruction's 4
    const int zc_l = (lane & 15) ^ (zpx << 1);

    auto stage = [&](int c, int buf) {
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(lds)) + (unsigned)buf * STAGE_B;
        const int seg = c % p.nseg;
        const int t = c / p.nseg;
        const int rg = t % p.nrg, b = t / p.nrg;
        const int h0 = rg * R, w0 = seg * N16_P;
#pragma unroll
        for (int i = 0; i < (NXI + NW - 1) / NW; ++i) {
            const int j = wave_u + NW * i;
            if (NXI % NW == 0 || j < NXI) {
                const int r = j / 5, part = j - r * 5;
                const int gr = h0 * SH - 1 + r;
                const int col = part * 8 + xpx;                  // 0..39 (34 used); bit 3 of col = bit 0 of part
                const int chunk = xc_l ^ ((part & 1) << 2);
                int gc = w0 - 1 + col;
                if (p.circ) {
                    if (gc < 0) gc += p.W;
                    else if (gc == p.W) gc = 0;
                }
                const bool ok = ci0 + chunk * 8 < p.Cin && col < N16_P + 2 && gc >= 0 && gc < p.W && gr >= 0 && gr < p.H;
                const unsigned soff = (gr >= 0 && gr < p.H) ? (unsigned)(((size_t)b * p.H + gr) * p.W * p.Cin * 2u) : 0u;
                dma16(x_rs, base + (unsigned)j * 1024u, ok ? (unsigned)(gc * p.Cin + ci0 + chunk * 8) * 2u : OOR, soff);
            }
        }
        static_assert(NZI % NW == 0, "dZ instructions split evenly over the waves");
#pragma unroll
        for (int i = 0; i < NZI / NW; ++i) {
            const int j = wave_u + NW * i;
            const int r = j >> 3, c4 = (j & 7) * 4;              // first pixel of the instruction: row r, column c4; bit 3 of the column = bit 1 of (j & 7)
            const int chunk = zc_l ^ ((((j & 7) >> 1) & 1) << 3);
            const int h = h0 + r, w = w0 + c4;
            const bool ok = co0 + chunk * 8 < p.Cout && h < p.Ho && w + zpx < p.Wo;
            const unsigned soff = (h < p.Ho && w < p.Wo) ? (unsigned)((((size_t)b * p.Ho + h) * p.Wo + w) * p.Cout * 2u) : 0u;
            dma16(z_rs, base + (unsigned)(X_B + j * 1024), ok ? (unsigned)(zpx * p.Cout + co0 + chunk * 8) * 2u : OOR, soff);
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) acc[t][bt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane roles: G = lane >> 4 = k-group (columns 8 G ..), q = (lane >> 2) & 3: the pixel of the 4-pixel block whose
    // address this lane supplies, pp = lane & 3: which 8 bytes of the block's 32
    const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    unsigned a_lane[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = 8 * G + 4 * h + q + kw;
            const int chunk = (2 * wm + (pp >> 1)) ^ ((((col >> 1) & 1) << 1) | (((col >> 3) & 1) << 2));
            a_lane[kw][h] = (unsigned)(col * 128 + chunk * 16 + 8 * (pp & 1));
        }
    unsigned b_lane[4];
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
        const int chunk = (8 * wn + 2 * bt + (pp >> 1)) ^ ((q << 1) | ((G & 1) << 3));
        b_lane[bt] = (unsigned)(X_B + (8 * G + q) * 256 + chunk * 16 + 8 * (pp & 1));
    }

    // bias gradient db[co] = sum over pixels of dZ: the waves of the first ci tile that hold ci rows 0-15 add up the dZ fragments they
    // read anyway on the VALU (8 bf16 -> fp32 adds per fragment: 32 per 36 MFMAs on a quarter of the waves of 1 / tiles_ci of the
    // workgroups) -- one register per co block instead of an accumulator tile each; the four k-groups meet at the end
    const bool do_bias = p.bias_part != nullptr && ci0 == 0 && wm == 0;     // wave-uniform
    float accb[4] = {0.f, 0.f, 0.f, 0.f};

    auto tr = [&](const unsigned char* sbase, unsigned off) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sbase + off));
    };
    auto bsum = [&](bf16x8 v) -> float {
        const u32x4 u = __builtin_bit_cast(u32x4, v);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += __builtin_bit_cast(float, u[i] << 16) + __builtin_bit_cast(float, u[i] & 0xffff0000u);
        return s;
    };
    auto compute = [&](int buf) {
        const unsigned char* sb = lds + buf * STAGE_B;
        bf16x8 fx[XR][3];
        auto load_row = [&](int j) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const s16x4 lo = tr(sb, a_lane[kw][0] + (unsigned)(j * N16_XP * 128));
                const s16x4 hi = tr(sb, a_lane[kw][1] + (unsigned)(j * N16_XP * 128));
                fx[j][kw] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        };
        auto load_b = [&](int r, int bt) -> bf16x8 {
            const s16x4 lo = tr(sb, b_lane[bt] + (unsigned)(r * N16_P * 256));
            const s16x4 hi = tr(sb, b_lane[bt] + (unsigned)((r * N16_P + 4) * 256));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
#pragma unroll
        for (int j = 0; j < 3; ++j) load_row(j);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // two co blocks at a time: their dZ fragments, 18 MFMAs; the next output row's halo row is requested under the second half
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8 f0 = load_b(r, 2 * half), f1 = load_b(r, 2 * half + 1);
                if (half == 1 && r + 1 < R) {
#pragma unroll
                    for (int j = r * SH + 3; j < (r + 1) * SH + 3; ++j) load_row(j);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    acc[t][2 * half] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[r * SH + t / 3][t % 3], f0, acc[t][2 * half], 0, 0, 0);
                    acc[t][2 * half + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[r * SH + t / 3][t % 3], f1, acc[t][2 * half + 1], 0, 0, 0);
                }
                if (do_bias) {
                    accb[2 * half] += bsum(f0);
                    accb[2 * half + 1] += bsum(f1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (c_begin < c_end) {
        stage(c_begin, 0);
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            wait_vmcnt<0>();                         // this wave's DMA of chunk c has landed
            __syncthreads();                         // ... everyone's has, and everyone is done reading the other buffer
            if (c + 1 < c_end) stage(c + 1, buf ^ 1);
            compute(buf);
        }
    }

    // ---- partial tile -> workspace [split][tap][ci][co]; 16x16 result: register r of lane l = (ci 4 (l >> 4) + r, co l & 15)
    float* out = p.ws + (size_t)split * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            const int co = co0 + wn * 64 + bt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wm * 16 + 4 * (lane >> 4) + r;
                if (ci < p.Cin && co < p.Cout) out[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][bt][r];
            }
        }
    if (do_bias) {                       // lanes l, l + 16, l + 32, l + 48 hold the four k-groups' sums of co (l & 15): fixed-order add
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            float v = accb[bt];
            v += __shfl_down(v, 32, 64);
            v += __shfl_down(v, 16, 64);
            const int co = co0 + wn * 64 + bt * 16 + lane;
            if (lane < 16 && co < p.Cout) p.bias_part[(size_t)split * p.Cout + co] = v;
        }
    }
}

// Which layers CAN take the 16x16x32 kernel: stride 1, full wave roles (Cin > 32, Cout > 64), whole 32-column segments. OFF by
// default: measured on the bench shapes (tools/bench_wgrad_bf16.py, B = 128, same box, alternating runs) it is 3-9 % SLOWER than the
// 32x32x16 kernel (layers 17 / 19 / 21: 0.334 / 0.658 / 0.601 ms against 0.323 / 0.651 / 0.550 ms incl. the reduction) -- the clock
// gain of the shape (MI355X_MICROARCH.md, DVFS give-back (7)) does not cover its 36 instead of 9 MFMA issues per k-step and the
// four dZ fragments per k-step. witw_conv3x3_wgrad_bf16_mfma16(1) / WITW_WGRAD16=1 select it (parity-tested like the default).
int g_wgrad16 = -1;
int wgrad16_enabled() {
    if (g_wgrad16 < 0) {
        const char* e = getenv("WITW_WGRAD16");
        g_wgrad16 = e ? (atoi(e) != 0) : 0;
    }
    return g_wgrad16;
}
bool wgrad_nh16_applies(int Wo, int Cin, int Cout, int stride_h) {
    // stride (2,1): its 9-row halo keeps 60 fragment registers live beside the 144 accumulators: 220 bytes of scratch per lane, not instantiated
    return wgrad16_enabled() && stride_h == 1 && (Wo % N16_P) == 0 && Cin > 32 && Cout > 64;
}

// wave roles for a layer (see the kernel): slabs that exist, the rest of the 8 waves along k
void wgrad_nh_roles(int Cin, int Cout, int stride_h, int* nwm, int* nwn) {
    *nwm = Cin <= 32 ? 1 : 2;
    *nwn = Cout <= 32 ? 1 : Cout <= 64 ? 2 : 4;
    if (stride_h == 2) *nwm = 2;         // instantiated at stride (2,1): (2,4), (2,2), (2,1) -- its layers have wide inputs
}
int wgrad_nh_ksplit(int Cin, int Cout, int stride_h) {
    int nwm, nwn;
    wgrad_nh_roles(Cin, Cout, stride_h, &nwm, &nwn);
    return 8 / (nwm * nwn);
}

// stage shape: R output rows per stage (8 rows x 16 columns of one image at stride 1: 8 k-steps = 72 MFMAs per wave between two
// barriers; 4 at stride (2,1), whose halo is 9 rows). Two stages. Measured at the bench shapes (B = 128, layers 17-21, tools/
// bench_wgrad_bf16.py, round 5): R = 4 with a ring of 3 or 4 stages was 7-9 % SLOWER than R = 8 with 2 (more barriers and halo
// re-reads; the DMA latency is already covered by one 72-MFMA phase), so the ring stays generic in the kernel but only NS = 2
// is instantiated.
int wgrad_nh_rows(int stride_h) { return stride_h == 2 ? 4 : 8; }

int wgrad_nh_splits(int B, int Ho, int Wo, int Cin, int Cout, int stride_h) {
    const int tiles = cdiv(Cin, WB_TM) * cdiv(Cout, WB_TN);
    const bool k16 = wgrad_nh16_applies(Wo, Cin, Cout, stride_h);      // stage = 4 rows x 32 columns instead of 8 x 16
    const int chunks = k16 ? B * cdiv(Ho, 4) * cdiv(Wo, N16_P) : B * cdiv(Ho, wgrad_nh_rows(stride_h)) * cdiv(Wo, NH_P);
    int splits = cdiv(witw_cu_count(), tiles);       // one workgroup per CU; every extra split costs a 36*Cin*Cout-byte partial
    if (splits > chunks) splits = chunks;
    if (splits < 1) splits = 1;
    return splits;
}

int wgrad_bf16_rows(int stride_h) { return stride_h == 2 ? 1 : 2; }

int wgrad_bf16_splits(int B8, int Ho, int Wo, int Cin, int Cout, int stride_h) {
    const int tiles = cdiv(Cin, WB_TM) * cdiv(Cout, WB_TN);
    const int chunks = B8 * cdiv(Ho, wgrad_bf16_rows(stride_h)) * cdiv(Wo, WB_P);
    int splits = cdiv(256, tiles);            // one workgroup per CU; every extra split costs a 36*Cin*Cout-byte partial
    if (splits > chunks) splits = chunks;
    if (splits < 1) splits = 1;
    return splits;
}

}  // namespace

extern "C" {

long long witw_octet_elems(int B, int H, int W, int C) { return (long long)cdiv(B, 8) * 8 * H * W * C; }

// x NHWC bf16 [B,H,W,C] (C % 8 == 0) -> y batch-octet bf16 [ceil(B/8)][H][W][C][8]
int witw_nhwc_bf16_to_octet(const void* x_bf16, void* y_bf16, int B, int H, int W, int C, void* stream) {
    WITW_CHECK_ARG(x_bf16 && y_bf16, "nhwc_bf16_to_octet: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "nhwc_bf16_to_octet: bad shape (C=%d must be a multiple of 8)", C);
    const size_t total = (size_t)cdiv(B, 8) * H * W * (C / 8);
    hipLaunchKernelGGL(nhwc_to_octet_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x_bf16, (unsigned short*)y_bf16, B, (size_t)H * W, C, total);
    WITW_CHECK_LAUNCH("nhwc_bf16_to_octet");
    return WITW_OK;
}

long long witw_conv3x3_wgrad_bf16_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h) {
    const int Ho = (H + 2 - 3) / stride_h + 1;
    const long long splits = wgrad_bf16_splits(cdiv(B, 8), Ho, W, Cin, Cout, stride_h);
    return splits * 9 * Cin * Cout + splits * Cout;
}

// x_oct [B8][H][W][Cin][8], dz_oct [B8][Ho][W][Cout][8] (batch-octet bf16, witw_nhwc_bf16_to_octet).
// dw [Cout][cin_real][3][3] fp32 (torch layout), db [Cout] fp32 or NULL. accumulate != 0 adds instead of overwriting.
int witw_conv3x3_wgrad_bf16(const void* x_oct, const void* dz_oct, float* dw, float* db, float* workspace, int B, int H, int W,
                            int Cin, int cin_real, int Cout, int stride_h, int pad_circular, int accumulate, void* stream) {
    WITW_CHECK_ARG(x_oct && dz_oct && dw && workspace, "conv3x3_wgrad_bf16: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad_bf16: bad shape");
    WITW_CHECK_ARG((Cin % 8) == 0 && (Cout % 8) == 0, "conv3x3_wgrad_bf16: Cin=%d and Cout=%d must be multiples of 8", Cin, Cout);
    WITW_CHECK_ARG(cin_real > 0 && cin_real <= Cin, "conv3x3_wgrad_bf16: cin_real=%d outside (0,%d]", cin_real, Cin);
    WITW_CHECK_ARG(stride_h == 1 || stride_h == 2, "conv3x3_wgrad_bf16: stride_h=%d unsupported", stride_h);
    const int B8 = cdiv(B, 8);
    const int Ho = (H + 2 - 3) / stride_h + 1;
    WITW_CHECK_ARG((size_t)B8 * H * W * Cin * 16 < 0x80000000ull && (size_t)B8 * Ho * W * Cout * 16 < 0x80000000ull,
                   "conv3x3_wgrad_bf16: operand too large for one buffer descriptor");
    hipStream_t st = (hipStream_t)stream;
    WgradBfArgs a;
    a.x = (const unsigned short*)x_oct; a.dz = (const unsigned short*)dz_oct; a.ws = workspace;
    a.B8 = B8; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Ho = Ho; a.Wo = W;
    a.circ = pad_circular;
    const int R = wgrad_bf16_rows(stride_h);
    a.nseg = cdiv(a.Wo, WB_P);
    a.nrg = cdiv(Ho, R);
    a.chunks = B8 * a.nrg * a.nseg;
    const int splits = wgrad_bf16_splits(B8, Ho, a.Wo, Cin, Cout, stride_h);
    a.cps = cdiv(a.chunks, splits);
    const size_t n = (size_t)9 * Cin * Cout;
    a.bias_part = db ? workspace + (size_t)splits * n : nullptr;
    const dim3 grid(cdiv(Cin, WB_TM), cdiv(Cout, WB_TN), splits);
    if (stride_h == 2)
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<2, 1>), grid, dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<1, 2>), grid, dim3(512), 0, st, a);
    WITW_CHECK_LAUNCH("conv3x3_wgrad_bf16");
    // (a split whose chunk range came out empty has written zero partials)
    hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3((unsigned)((n + Cout + 255) / 256)), dim3(256), 0, st, workspace, dw, Cin,
                       Cout, splits, accumulate, cin_real, a.bias_part, db);
    WITW_CHECK_LAUNCH("wgrad_bf16_reduce");
    return WITW_OK;
}


// 1: stride-1 layers with Cin > 32, Cout > 64 and W % 32 == 0 run the NHWC weight gradient on v_mfma_f32_16x16x32_bf16
// (conv3x3_wgrad_bf16_nhwc16_kernel); 0 (default: it measured slower, see above): on 32x32x16. enable < 0 only queries. Returns the
// previous setting. Call it before sizing the workspace: the two forms split K differently.
int witw_conv3x3_wgrad_bf16_mfma16(int enable) {
    const int prev = wgrad16_enabled();
    if (enable >= 0) g_wgrad16 = enable != 0;
    return prev;
}

long long witw_conv3x3_wgrad_bf16_nhwc_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h) {
    const int Ho = (H + 2 - 3) / stride_h + 1;
    const long long splits = wgrad_nh_splits(B, Ho, W, Cin, Cout, stride_h);
    return splits * 9 * Cin * Cout + splits * Cout;
}

// The same gradient from the NHWC tensors themselves: x [B][H][W][Cin], dz [B][Ho][W][Cout] bf16 (what the bf16 forward and
// dgrad launches write) -- no re-layout pass. dw [Cout][cin_real][3][3] fp32, db [Cout] fp32 or NULL; workspace of
// witw_conv3x3_wgrad_bf16_nhwc_workspace_floats floats. Bitwise reproducible (fixed split order), not bit-equal to the octet
// entry (different summation order within a split).
int witw_conv3x3_wgrad_bf16_nhwc(const void* x_nhwc, const void* dz_nhwc, float* dw, float* db, float* workspace, int B, int H, int W,
                                 int Cin, int cin_real, int Cout, int stride_h, int pad_circular, int accumulate, void* stream) {
    WITW_CHECK_ARG(x_nhwc && dz_nhwc && dw && workspace, "conv3x3_wgrad_bf16_nhwc: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad_bf16_nhwc: bad shape");
    WITW_CHECK_ARG((Cin % 8) == 0 && (Cout % 8) == 0, "conv3x3_wgrad_bf16_nhwc: Cin=%d and Cout=%d must be multiples of 8", Cin, Cout);
    WITW_CHECK_ARG(cin_real > 0 && cin_real <= Cin, "conv3x3_wgrad_bf16_nhwc: cin_real=%d outside (0,%d]", cin_real, Cin);
    WITW_CHECK_ARG(stride_h == 1 || stride_h == 2, "conv3x3_wgrad_bf16_nhwc: stride_h=%d unsupported", stride_h);
    const int Ho = (H + 2 - 3) / stride_h + 1;
    WITW_CHECK_ARG((size_t)B * H * W * Cin * 2 < 0x80000000ull && (size_t)B * Ho * W * Cout * 2 < 0x80000000ull,
                   "conv3x3_wgrad_bf16_nhwc: operand too large for one buffer descriptor");
    hipStream_t st = (hipStream_t)stream;
    WgradNhArgs a;
    a.x = (const unsigned short*)x_nhwc; a.dz = (const unsigned short*)dz_nhwc; a.ws = workspace;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Ho = Ho; a.Wo = W;
    a.circ = pad_circular;
    const bool k16 = wgrad_nh16_applies(a.Wo, Cin, Cout, stride_h);
    const int R = k16 ? 4 : wgrad_nh_rows(stride_h);
    a.nseg = cdiv(a.Wo, k16 ? N16_P : NH_P);
    a.nrg = cdiv(Ho, R);
    a.chunks = B * a.nrg * a.nseg;
    a.tiles_ci = cdiv(Cin, WB_TM); a.tiles_co = cdiv(Cout, WB_TN);
    a.splits = wgrad_nh_splits(B, Ho, a.Wo, Cin, Cout, stride_h);
    a.cps = cdiv(a.chunks, a.splits);
    const size_t n = (size_t)9 * Cin * Cout;
    int nwm, nwn;
    wgrad_nh_roles(Cin, Cout, stride_h, &nwm, &nwn);
    const int parts = a.splits;                          // partial tiles in the workspace
    a.bias_part = db ? workspace + (size_t)parts * n : nullptr;
    const dim3 grid((unsigned)(a.tiles_ci * a.tiles_co * a.splits));
#define WITW_NH_LAUNCH(SH_, R_, NWM_, NWN_) \
    hipLaunchKernelGGL((conv3x3_wgrad_bf16_nhwc_kernel<SH_, R_, 2, NWM_, NWN_>), grid, dim3(512), 0, st, a)
    if (k16) {
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_nhwc16_kernel<1, 4>), grid, dim3(512), 0, st, a);
    } else if (stride_h == 2) {
        if (nwn == 4) WITW_NH_LAUNCH(2, 4, 2, 4);
        else if (nwn == 2) WITW_NH_LAUNCH(2, 4, 2, 2);
        else WITW_NH_LAUNCH(2, 4, 2, 1);
    } else if (nwm == 2) {
        if (nwn == 4) WITW_NH_LAUNCH(1, 8, 2, 4);
        else if (nwn == 2) WITW_NH_LAUNCH(1, 8, 2, 2);
        else WITW_NH_LAUNCH(1, 8, 2, 1);
    } else {
        if (nwn == 4) WITW_NH_LAUNCH(1, 8, 1, 4);
        else if (nwn == 2) WITW_NH_LAUNCH(1, 8, 1, 2);
        else WITW_NH_LAUNCH(1, 8, 1, 1);
    }
#undef WITW_NH_LAUNCH
    WITW_CHECK_LAUNCH("conv3x3_wgrad_bf16_nhwc");
    hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3((unsigned)((n + Cout + 255) / 256)), dim3(256), 0, st, workspace, dw, Cin,
                       Cout, parts, accumulate, cin_real, a.bias_part, db);
    WITW_CHECK_LAUNCH("wgrad_bf16_reduce");
    if (k16) witw_note_variant("conv3x3_wgrad_bf16_nhwc16_kernel<%d,%d>", stride_h, R);
    else witw_note_variant("conv3x3_wgrad_bf16_nhwc_kernel<%d,%d,%d,%d>", stride_h, R, nwm, nwn);
    return WITW_OK;
}

}  // extern "C"
