// 3x3 convolution, NHWC fp32, implicit GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces, for the FOV_DSM encoder of the reference (model/cvig_fov.py:256-294):
//   torch.nn.Conv2d(k=3, pad=1, stride (1,1)|(2,1)) [+ HorizCircPadding :212-231]
//   [+ AddDropout :234-245] [+ ReLU] [+ MaxPool2d(2,2)]
// as ONE kernel per conv layer: padding is a halo-load policy (zero rows, zero or
// wrapped columns), bias / dropout-scale / ReLU / 2x2 max-pool live in the epilogue.
//
// Tiling (per workgroup of NW = 4 or 8 waves; 8 = two waves per SIMD sharing one weight slab):
//   output tile  TH x TW = NW x 64 conv-output pixels  (2*NW MFMA M-tiles of 32 columns)
//   x TN output channels (64 or 128)                  (2 or 4 MFMA N-tiles of 32)
//   K loop over input channels in chunks of 8 (two "quads" of 4 channels); per chunk
//   the (TH-1)*SH+3 x 66 input halo tile and the 9x8xTN weight slab are staged in LDS
//   (double buffered, register-staged one chunk ahead of the MFMAs).
// LDS images are quad-planar ([quad][row][col] of float4 / [tap][quad][n] of float4)
// so every ds_read_b128 of a wave covers 2 x 512 contiguous bytes (conflict free) and
// yields the operands of 4 MFMA k-steps: MFMA step j takes channel j of quad 0 from
// lanes 0-31 and channel j of quad 1 from lanes 32-63 (A and B agree on that order).
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int TW_WIDE = 64;

// Source of every padded (out-of-image) halo slot: loading zeros from memory keeps the staging
// path free of selects, which the compiler would otherwise pin right behind each load (vmcnt(0)).
__device__ const f32x4 g_zero_f4[1] = {{0.f, 0.f, 0.f, 0.f}};

struct ConvArgs {
    const float* x;         // [B,H,W,Cin] NHWC, Cin % 8 == 0
    const float* wpk;       // packed: [n_tile][cin/8][tap][quad][TN][4]
    const float* bias;      // [n_tiles*TN] (zero padded)
    const float* dropmask;  // [B,Cout] scale (0 or 1/(1-p)) or nullptr
    const float* post_scale;  // nullptr, or per-channel scale/shift applied AFTER the activation (folded BatchNorm)
    const float* post_shift;
    float lrelu;            // negative slope when relu == 2 (LeakyReLU)
    unsigned char* pool_code;  // nullptr, or [B,Hy,Wy,Cout] argmax position (dy*2+dx) of the fused 2x2 max pool
    const float* gate;      // nullptr, or a tensor shaped like y: outputs where gate <= 0 are zeroed (ReLU backward)
    float* y;               // NHWC [B,Hy,Wy,Cout] or NCHW [B,Cout,Hy,Wy]
    int B, H, W, Cin, Cout;
    int Ho, Wo;             // conv output size (before pooling)
    int tiles_x, tiles_y;
    int n_tiles, sp_total, sp_per_xcd, xcd_map;   // workgroup -> tile mapping, see the kernel
    int circ, relu, out_nchw;
    int force_nw;           // 0 = choose, 4 / 8 = force the workgroup shape (tuning aid)
    int force_geo;          // -1 = choose, 0 / 1 = force the wide / narrow tile geometry
    int dil_h;              // 1: input rows are zero-interleaved (row 2i = physical row i): dgrad of a stride-(2,1) conv
    int tap_base;           // TAPS == 4 only: the 2x2 taps are (tap_base + {0,1}, tap_base + {0,1}) of the 3x3 window
    // TAPS == 4 only, split-K: ksplit > 1 launches grid.y = ksplit workgroups per tile, each reducing K-chunks
    // [blockIdx.y * kc_per_split, ...) and writing its RAW partial sums (no bias / activation / affine) to
    // y + blockIdx.y * split_stride; splitk_finish_kernel adds the partials in fixed order and applies the epilogue.
    int ksplit, kc_per_split;
    size_t split_stride;
    // TAPS == 4 only: s2d != 0 writes the output as the space-to-depth(2) image the next k=4,s=2 layer reads,
    // [B, s2d_h, s2d_w, 4*Cout] with channel ((y&1)*2 + (x&1))*Cout + c; outputs at y >= valid_h or x >= valid_w are
    // written as 0 (the conv grid is padded to the tile; the next layer's window must read zeros there).
    int s2d, s2d_h, s2d_w, valid_h, valid_w;
#ifdef WITW_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/conv_stamps.cpp)
#endif
};

// NW = waves per workgroup (4: 4x64-pixel tile, one wave per SIMD; 8: 8x64-pixel tile, two waves per SIMD
// sharing one weight slab: 40 % less staging and half the barriers per MFMA, used when the layer is tall
// and the grid large enough).
// GEO = shape of one MFMA M-tile (32 output pixels): 0 = 1 row x 32 columns, workgroup tile NW rows x 64 columns
// (wide maps); 1 = 2 rows x 16 columns, workgroup tile 4*NW rows x 16 columns (maps up to 32 columns wide, e.g. the
// fov-70 ground branch whose widths are 24 and 12 in the deep layers: a 64-column tile would idle 63-81 % of it).
// GEO = 2 (round 5) = geometry 0 for a ZERO-INTERLEAVED input (dil_h: the dgrad of a stride-(2,1) layer, model/cvig_fov.py:263-272): every
// odd logical input row is zero, so an M-tile on an even output row only meets data under tap row kh = 1 and one on an odd output
// row under kh = 0 and 2 -- the other (M-tile, tap) products multiply staged zeros. This instantiation issues neither those MFMAs nor
// their fragment reads: half the matrix work of the launch, the same sums bit for bit (a skipped product adds +-0). 8-wave 128-channel
// tile only (its M-tiles are whole rows of known parity: row 2 wm + (mt >> 1) of an 8-row tile).
// TAPS = 9: the full 3x3 window. TAPS = 4: a 2x2 sub-window of it (rows/columns tap_base + {0,1}): the cvig_baseline
// Conv2d(k=4, s=2, p=0) (model/cvig_baseline.py:236-252) is a 2x2 convolution over the space-to-depth(2) image, i.e. the
// 3x3 window whose first tap row and column are zero (dgrad: last row and column) — 4 of 9 taps carry all the work.
template <int TN, int SH, bool POOL, int NW, int GEO, int TAPS = 9>
__global__ __launch_bounds__(64 * NW) void conv3x3_nhwc_f32_kernel(ConvArgs p) {
    static_assert(TAPS == 9 || TAPS == 4, "3x3 window or a 2x2 sub-window");
    constexpr bool NARROW = GEO == 1;
    constexpr bool DIL = GEO == 2;              // zero-interleaved input rows: dead (M-tile, tap) products are not issued
    static_assert(!DIL || (TN == 128 && NW == 8 && SH == 1 && TAPS == 9 && !POOL), "the dilated form exists for the 8-wave 128-channel tile");
    constexpr int TH = NARROW ? 4 * NW : NW;
    constexpr int TW = NARROW ? 16 : 64;
    constexpr int IW = TW + 2;
    constexpr int NTHREADS = 64 * NW;
    constexpr int IH = (TH - 1) * SH + 3;
    constexpr int IN_F4 = 2 * IH * IW;          // float4 slots of one input stage
    constexpr int W_F4 = TAPS * 2 * TN;         // float4 slots of one weight stage
    constexpr int STAGE_F4 = IN_F4 + W_F4;
    constexpr int NIN = (IN_F4 + NTHREADS - 1) / NTHREADS;
    constexpr int NWT = (W_F4 + NTHREADS - 1) / NTHREADS;
    constexpr int WGM = (TN == 128) ? NW / 2 : NW;   // waves along M
    constexpr int WM = (2 * NW) / WGM;               // M-tiles per wave (2*NW per workgroup): 4 at TN=128, 2 at TN=64
    constexpr int WN = 2;                       // N-tiles per wave (64 channels)

    // +1: dummy slot that absorbs out-of-tile staging stores. The epilogue re-uses the buffer as NW wave-private 8 KB slabs:
    // with 4 taps and TN = 64 two stages are smaller than 8 slabs (58.6 KB < 64 KB), so the array is sized for both uses.
    constexpr int SMEM_F4 = (2 * STAGE_F4 + 1 > NW * 512) ? 2 * STAGE_F4 + 1 : NW * 512;
    __shared__ f32x4 smem[SMEM_F4];
    static_assert(SMEM_F4 * 16 >= NW * 32 * 64 * 4, "the buffer must hold one epilogue slab per wave");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hq = lane >> 5;

    // ---- block -> (n tile, image, spatial tile). Workgroups go to the 8 XCDs round-robin (block i -> XCD i % 8), each
    // with its own L2. XCD x takes the contiguous range [x*sp_per_xcd, (x+1)*sp_per_xcd) of spatial tiles and runs
    // the n tiles of one spatial tile back to back: the halo tile is fetched over the fabric once instead of once per
    // n tile, vertically adjacent tiles share their halo rows in L2, and the (small) weight chunks of all n tiles stay
    // L2 resident. The grid is padded to 8 * sp_per_xcd * n_tiles; surplus workgroups leave here.
    int ntile, sp;
    if (p.xcd_map) {
        const int g = blockIdx.x >> 3;
        ntile = g % p.n_tiles;
        sp = (blockIdx.x & 7) * p.sp_per_xcd + g / p.n_tiles;
        if (sp >= p.sp_total) return;
    } else {        // n tile slowest: the workgroups in flight share one weight slab
        ntile = blockIdx.x / p.sp_total;
        sp = blockIdx.x - ntile * p.sp_total;
    }
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int b = sp / tiles_img;
    sp -= b * tiles_img;
    const int ty = sp / p.tiles_x;
    const int tx = sp - ty * p.tiles_x;
    const int oy0 = ty * TH;
    const int ox0 = tx * TW;
    const int n0 = ntile * TN;
    int nkc = p.Cin >> 3;      // K-chunks (8 input channels each) this workgroup reduces
    int kc_lo = 0;             // ... starting at this one: split-K slices move the buffer bases, the loop stays 0 .. nkc
    if constexpr (TAPS == 4) {
        if (p.ksplit > 1) {
            kc_lo = (int)blockIdx.y * p.kc_per_split;
            nkc = min(nkc - kc_lo, p.kc_per_split);
        }
    }

    // ---- staging through buffer loads: one descriptor per image / per weight tile built from scalars, the
    // per-thread byte offset is fixed for the whole K loop and the K-chunk advance rides in the scalar
    // offset, so a load costs no VALU instruction next to the f32 MFMAs. Padded halo slots (rows outside the
    // image, zero-padded columns) carry an out-of-range offset and read 0.
    constexpr unsigned OOR = 0x80000000u;
    const int Hp = p.dil_h ? (p.H - 1) / 2 + 1 : p.H;          // physical rows of the input
    const size_t img_floats = (size_t)Hp * p.W * p.Cin;
    __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)b * img_floats + (size_t)kc_lo * 8), 0, (unsigned)(img_floats * 4) - (unsigned)kc_lo * 32u, 0x00020000);
    const unsigned wtile_bytes = (unsigned)nkc * W_F4 * 16u;
    __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const f32x4*>(p.wpk) + ((size_t)ntile * (p.Cin >> 3) + kc_lo) * W_F4), 0, wtile_bytes, 0x00020000);
    unsigned gin[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int s = tid + i * NTHREADS;
        const int pix = s >> 1, q = s & 1;
        const int r = pix / IW, c = pix - r * IW;
        const int gr = oy0 * SH - 1 + r;
        int gc = ox0 - 1 + c;
        bool ok = (s < IN_F4) && gr >= 0 && gr < p.H;
        int grp = gr;                 // physical row
        if (p.dil_h) {
            ok = ok && (gr & 1) == 0;
            grp = gr >> 1;
        }
        if (p.circ) {
            gc %= p.W;
            if (gc < 0) gc += p.W;
        } else {
            ok = ok && gc >= 0 && gc < p.W;
        }
        gin[i] = ok ? (unsigned)((((size_t)grp * p.W + gc) * p.Cin + q * 4) * 4) : OOR;
    }
    const unsigned gwoff = (unsigned)tid * 16u;

    // Staging is branch-free so that the whole K-chunk body is ONE scheduling region; out-of-tile
    // slots are sent to the dummy LDS slot.
    f32x4 rin[NIN], rw[NWT];
    auto load_stage = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NIN; ++i)
            rin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, gin[i], (unsigned)kc * 32u, 0));
        const unsigned wbase = (unsigned)kc * W_F4 * 16u;
#pragma unroll
        for (int i = 0; i < NWT; ++i)   // slots past the slab read past the tile (still inside wpk or range-checked to 0)
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, gwoff, wbase + (unsigned)i * (NTHREADS * 16u), 0));
    };
    auto store_stage = [&](int buf) {
        f32x4* in_s = smem + buf * STAGE_F4;
        f32x4* w_s = in_s + IN_F4;
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int s = tid + i * NTHREADS;
            f32x4* dst = (NIN * NTHREADS == IN_F4 || s < IN_F4) ? in_s + (s & 1) * (IH * IW) + (s >> 1) : smem + 2 * STAGE_F4;
            *dst = rin[i];
        }
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int s = tid + i * NTHREADS;
            f32x4* dst = (NWT * NTHREADS == W_F4 || s < W_F4) ? w_s + s : smem + 2 * STAGE_F4;
            *dst = rw[i];
        }
    };

    // ---- wave -> M-tiles / N-tiles
    const int wm = (TN == 128) ? (wave >> 1) : wave;
    const int wn = (TN == 128) ? (wave & 1) : 0;
    int trow[WM], tcol[WM];  // first tile row / column of each M-tile of this wave
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        if (NARROW) {                    // M-tile t covers tile rows 2t, 2t+1 and all 16 columns
            trow[mt] = 2 * ((TN == 128 ? 4 : 2) * wm + mt);
            tcol[mt] = 0;
        } else if (TN == 128) {
            trow[mt] = 2 * wm + (mt >> 1);
            tcol[mt] = 32 * (mt & 1);
        } else {
            trow[mt] = 2 * (wm >> 1) + mt;
            tcol[mt] = 32 * (wm & 1);
        }
    }
    // pixel m (0..31) of an M-tile -> (row, column) offset inside it
    auto m_row = [](int m) { return NARROW ? (m >> 4) : 0; };
    auto m_col = [](int m) { return NARROW ? (m & 15) : m; };
    int abase[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
        abase[mt] = hq * (IH * IW) + (trow[mt] + m_row(l31)) * SH * IW + tcol[mt] + m_col(l31) + (TAPS == 4 ? p.tap_base * (IW + 1) : 0);
    const int wbase = hq * TN + wn * 64 + l31;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- main loop. Per K chunk: 9 taps x 4 k-steps x (WM x WN) MFMAs. The operand fragments of
    // tap t+1 are read from LDS BEFORE the MFMAs of tap t are issued (two register sets), the
    // next chunk's global loads are issued at the top of the chunk and written to the other LDS
    // buffer after tap 4, and the chunk's single barrier sits in front of tap 8 so that the
    // first fragments of the next chunk are fetched behind tap 8's MFMAs: the matrix pipe only
    // idles for the barrier skew.
    f32x4 fa[2][WM], fb[2][WN];
    // DIL: M-tile mt sits on output row 2 wm + (mt >> 1) of an 8-row tile at an even oy0: its input row under tap row kh is real
    // (even) iff (mt >> 1) + kh is odd
    auto live = [](int mt, int tap) { return !DIL || ((((mt >> 1) + tap / 3) & 1) == 1); };
    auto read_frags = [&](int set, const f32x4* in_s, const f32x4* w_s, int tap) {
        const int kh = (TAPS == 4) ? (tap >> 1) : tap / 3, kw = (TAPS == 4) ? (tap & 1) : tap - kh * 3;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
            if (live(mt, tap)) fa[set][mt] = in_s[abase[mt] + kh * IW + kw];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) fb[set][nt] = w_s[tap * 2 * TN + wbase + nt * 32];
    };
    auto mfma_tap = [&](int set, int tap) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
                if (live(mt, tap)) {
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt)
                        acc[mt][nt] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][mt][j], fb[set][nt][j], acc[mt][nt], 0, 0, 0);
                }
    };

#ifdef WITW_STAMPS
    unsigned long long t_stamp[6];
    int n_stamp = 0;
#define STAMP() do { if (n_stamp < 6) t_stamp[n_stamp++] = __builtin_amdgcn_s_memtime(); } while (0)
    const unsigned long long r_stamp0 = __builtin_amdgcn_s_memrealtime();
#else
#define STAMP() do { } while (0)
#endif
    STAMP();
    load_stage(0);
    store_stage(0);
    __syncthreads();
    read_frags(0, smem, smem + IN_F4, 0);
    STAMP();

    // Issue-order recipe for one tap (LLVM sched groups: 0x8 MFMA, 0x20 VMEM read, 0x100 DS read,
    // 0x200 DS write): every LDS / global instruction is issued alone between MFMAs so that it
    // hides in the shadow of a 64-cycle v_mfma_f32_32x32x2_f32 instead of stalling the pipe.
    constexpr int MPT = 4 * (DIL ? WM / 2 : WM) * WN;   // MFMAs per tap (DIL: half the M-tiles are live under any tap)
    constexpr int RPT = (DIL ? WM / 2 : WM) + WN;       // fragment reads per tap
#define SG_PLAIN_TAP()                                                  \
    do {                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < RPT; ++i_) {            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          \
        }                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < MPT - 2 * RPT; ++i_)    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
    } while (0)
#define SG_HEAVY_TAP(MASK, NX)                                          \
    do {                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < RPT; ++i_) {            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          \
        }                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < (NX); ++i_) {           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);           \
        }                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < MPT - RPT - (NX); ++i_) \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
    } while (0)
    static_assert(MPT - RPT - (NIN + NWT) >= 0, "tap too short to hide the staging instructions");

    if constexpr (TAPS == 4) {
        // 4 taps per chunk: global loads of the next chunk ride under tap 0, its LDS writes under tap 2, the chunk's
        // barrier sits in front of tap 3, behind whose MFMAs the first fragments of the next chunk arrive. An even
        // tap count keeps the fragment sets aligned from chunk to chunk (no register copy).
        for (int kc = 0; kc < nkc; ++kc) {
            const int cur = kc & 1;
            const int kn = (kc + 1 < nkc) ? kc + 1 : kc;
            const f32x4* in_s = smem + cur * STAGE_F4;
            const f32x4* w_s = in_s + IN_F4;
            const f32x4* in_n = smem + (cur ^ 1) * STAGE_F4;
            read_frags(1, in_s, w_s, 1);
            load_stage(kn);
            mfma_tap(0, 0);
            SG_HEAVY_TAP(0x020, NIN + NWT);
            read_frags(0, in_s, w_s, 2);
            mfma_tap(1, 1);
            SG_PLAIN_TAP();
            read_frags(1, in_s, w_s, 3);
            store_stage(cur ^ 1);
            mfma_tap(0, 2);
            SG_HEAVY_TAP(0x200, NIN + NWT);
            __syncthreads();
            read_frags(0, in_n, in_n + IN_F4, 0);
            mfma_tap(1, 3);
            SG_PLAIN_TAP();
        }
    } else {
        for (int kc = 0; kc < nkc; ++kc) {
            const int cur = kc & 1;
            const int kn = (kc + 1 < nkc) ? kc + 1 : kc;   // last chunk restages itself (never read)
            const f32x4* in_s = smem + cur * STAGE_F4;
            const f32x4* w_s = in_s + IN_F4;
            const f32x4* in_n = smem + (cur ^ 1) * STAGE_F4;
            // tap 0 (+ global loads of the next chunk)
            read_frags(1, in_s, w_s, 1);
#ifndef WITW_DIAG_NOSTAGE
            load_stage(kn);
#endif
            mfma_tap(0, 0);
            SG_HEAVY_TAP(0x020, NIN + NWT);
#pragma unroll
            for (int tap = 1; tap < 4; ++tap) {
                read_frags((tap + 1) & 1, in_s, w_s, tap + 1);
                mfma_tap(tap & 1, tap);
                SG_PLAIN_TAP();
            }
            // tap 4 (+ LDS writes of the next chunk)
            read_frags(1, in_s, w_s, 5);
#ifndef WITW_DIAG_NOSTAGE
            store_stage(cur ^ 1);
#endif
            mfma_tap(0, 4);
            SG_HEAVY_TAP(0x200, NIN + NWT);
#pragma unroll
            for (int tap = 5; tap < 8; ++tap) {
                read_frags((tap + 1) & 1, in_s, w_s, tap + 1);
                mfma_tap(tap & 1, tap);
                SG_PLAIN_TAP();
            }
#ifndef WITW_DIAG_NOBARRIER
            __syncthreads();
#endif
            // tap 8, behind which the first fragments of the next chunk arrive
            read_frags(1, in_n, in_n + IN_F4, 0);
            mfma_tap(0, 8);
            SG_PLAIN_TAP();
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) fa[0][mt] = fa[1][mt];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) fb[0][nt] = fb[1][nt];
        }
    }
#undef SG_PLAIN_TAP
#undef SG_HEAVY_TAP

    STAMP();
    // ---- epilogue: bias, dropout scale, ReLU, optional 2x2 max pool, store
    float bv[WN], dm[WN], ps[WN], pt[WN];
    int nch[WN];
    int act = p.relu;
    bool has_post = p.post_scale != nullptr;
    bool raw = false;          // split-K: the partial sums leave untouched (+0, *1, no activation), splitk_finish_kernel finishes
    if constexpr (TAPS == 4) {
        raw = p.ksplit > 1;
        if (raw) { act = 0; has_post = false; }
    }
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        nch[nt] = n0 + wn * 64 + nt * 32 + l31;
        bv[nt] = raw ? 0.f : p.bias[nch[nt]];
        dm[nt] = 1.f;
        if (p.dropmask != nullptr && nch[nt] < p.Cout) dm[nt] = p.dropmask[(size_t)b * p.Cout + nch[nt]];
        ps[nt] = 1.f;
        pt[nt] = 0.f;
        if (has_post && nch[nt] < p.Cout) {
            ps[nt] = p.post_scale[nch[nt]];
            pt[nt] = p.post_shift[nch[nt]];
        }
    }
    // conv + bias -> Dropout2d scale -> activation -> per-channel affine
    auto fin = [&](float v, int nt) {
        v = (v + bv[nt]) * dm[nt];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = v > 0.f ? v : v * p.lrelu;
        if (has_post) v = v * ps[nt] + pt[nt];
        return v;
    };
    const int Hy = POOL ? (p.Ho >> 1) : p.Ho;
    const int Wy = POOL ? (p.Wo >> 1) : p.Wo;

    auto emit = [&](float v, int nt, int yy, int xx) {
        v = fin(v, nt);
        if (yy < Hy && xx < Wy && nch[nt] < p.Cout) {
            size_t o;
            if (p.out_nchw)
                o = (((size_t)b * p.Cout + nch[nt]) * Hy + yy) * Wy + xx;
            else
                o = (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nch[nt];
            if (p.gate != nullptr && !(p.gate[o] > 0.f)) v = 0.f;
            p.y[o] = v;
        }
    };

    if (!POOL && !p.out_nchw && (p.Cout & 3) == 0) {
        // Wide NHWC store: one M-tile (32 pixels x this wave's 64 channels) at a time is transposed
        // through a wave-private 8 KB LDS slab (the staging buffers are dead: every useful fragment
        // read precedes the loop's last barrier) and leaves as 16-byte stores, 4 pixels x 256 B per
        // wave-instruction, instead of 4-byte stores straight from the accumulator layout.
        float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 64);
        const int prow = lane >> 4, pc4 = (lane & 15) * 4;   // read-back role: pixel row in a group of 4, channel quad
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fin(acc[mt][nt][r], nt);
                    slab[((r & 3) + 8 * (r >> 2) + 4 * hq) * 64 + nt * 32 + l31] = v;
                }
            const int nbase = n0 + wn * 64 + pc4;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int m = g * 4 + prow;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc4);
                const int yy = oy0 + trow[mt] + m_row(m);
                const int xx = ox0 + tcol[mt] + m_col(m);
                bool ok = yy < Hy && xx < Wy && nbase < p.Cout;
                size_t o = (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nbase;
                f32x4 w = v;
                if constexpr (TAPS == 4) {
                    if (p.ksplit > 1) {
                        o += (size_t)blockIdx.y * p.split_stride;
                    } else if (p.s2d) {
                        ok = yy < 2 * p.s2d_h && xx < 2 * p.s2d_w && nbase < p.Cout;
                        o = ((((size_t)b * p.s2d_h + (yy >> 1)) * p.s2d_w + (xx >> 1)) * 4 + (yy & 1) * 2 + (xx & 1)) * p.Cout + nbase;
                        if (!(yy < p.valid_h && xx < p.valid_w)) w = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (ok) {
                    if (p.gate != nullptr) {
                        const f32x4 gt = *reinterpret_cast<const f32x4*>(p.gate + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = gt[e] > 0.f ? v[e] : 0.f;
                    }
                    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p.y + o));
                }
            }
        }
    } else if (!POOL) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * hq;
                    emit(acc[mt][nt][r], nt, oy0 + trow[mt] + m_row(m), ox0 + tcol[mt] + m_col(m));
                }
    } else if (NARROW) {
        // Narrow geometry: both rows of a pooling window sit in ONE M-tile (registers r and r+8), its two columns in
        // registers r and r+1: 8 pooled pixels (1 row x 8 columns) per M-tile, all in-lane.
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            const int yy = (oy0 + trow[mt]) >> 1;
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const float a00 = acc[mt][nt][r], a01 = acc[mt][nt][r + 1];
                    const float a10 = acc[mt][nt][r + 8], a11 = acc[mt][nt][r + 9];
                    const float m = fmaxf(fmaxf(a00, a01), fmaxf(a10, a11));
                    const int xx = (ox0 >> 1) + (((r & 3) >> 1) | (hq << 1) | (((r >> 2) & 1) << 2));
                    emit(m, nt, yy, xx);
                    if (p.pool_code != nullptr && yy < Hy && xx < Wy && nch[nt] < p.Cout) {
                        const int code = (a00 == m) ? 0 : (a01 == m) ? 1 : (a10 == m) ? 2 : 3;
                        p.pool_code[(((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nch[nt]] = (unsigned char)code;
                    }
                }
        }
    } else if ((p.Cout & 3) == 0 && p.gate == nullptr) {
        // Fused 2x2 max-pool, wide store: rows (2a, 2a+1) of one column half sit in M-tiles (mtA, mtB) of this
        // wave; the 16 pooled pixels x 64 channels of the pair go through the wave-private LDS slab and leave
        // as 16-byte stores (4 pooled pixels x 256 B per wave-instruction).
        float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 64);
        const int prow = lane >> 4, pc4 = (lane & 15) * 4;
#pragma unroll
        for (int pr = 0; pr < WM / 2; ++pr) {
            const int mtA = (TN == 128) ? (pr & 1) : 0;       // TN==128: tiles {0,1}=row0 halves, {2,3}=row1
            const int mtB = (TN == 128) ? (2 + (pr & 1)) : 1;
            const int yy = (oy0 + trow[mtA]) >> 1;
            const int xb = (ox0 + tcol[mtA]) >> 1;
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float a00 = acc[mtA][nt][4 * g + 2 * e], a01 = acc[mtA][nt][4 * g + 2 * e + 1];
                        const float a10 = acc[mtB][nt][4 * g + 2 * e], a11 = acc[mtB][nt][4 * g + 2 * e + 1];
                        const float m = fmaxf(fmaxf(a00, a01), fmaxf(a10, a11));
                        const int pc = 4 * g + 2 * hq + e;
                        slab[pc * 64 + nt * 32 + l31] = fin(m, nt);
                        if (p.pool_code != nullptr && yy < Hy && xb + pc < Wy && nch[nt] < p.Cout) {
                            // first position attaining the max, scan order (0,0),(0,1),(1,0),(1,1) as torch's max_pool2d
                            const int code = (a00 == m) ? 0 : (a01 == m) ? 1 : (a10 == m) ? 2 : 3;
                            p.pool_code[(((size_t)b * Hy + yy) * Wy + xb + pc) * p.Cout + nch[nt]] = (unsigned char)code;
                        }
                    }
            const int nbase = n0 + wn * 64 + pc4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int pc = g * 4 + prow;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + pc * 64 + pc4);
                if (yy < Hy && xb + pc < Wy && nbase < p.Cout)
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.y + (((size_t)b * Hy + yy) * Wy + xb + pc) * p.Cout + nbase));
            }
        }
    } else {
        // rows (2a, 2a+1) of one column half sit in M-tiles (mtA, mtB) of this wave
#pragma unroll
        for (int pr = 0; pr < WM / 2; ++pr) {
            const int mtA = (TN == 128) ? (pr & 1) : 0;       // TN==128: tiles {0,1}=row0 halves, {2,3}=row1
            const int mtB = (TN == 128) ? (2 + (pr & 1)) : 1;
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float a00 = acc[mtA][nt][4 * g + 2 * e], a01 = acc[mtA][nt][4 * g + 2 * e + 1];
                        const float a10 = acc[mtB][nt][4 * g + 2 * e], a11 = acc[mtB][nt][4 * g + 2 * e + 1];
                        const float v0 = fmaxf(a00, a01);
                        const float v1 = fmaxf(a10, a11);
                        const int yy = (oy0 + trow[mtA]) >> 1;
                        const int xx = ((ox0 + tcol[mtA]) >> 1) + 4 * g + 2 * hq + e;
                        emit(fmaxf(v0, v1), nt, yy, xx);
                        if (p.pool_code != nullptr && yy < Hy && xx < Wy && nch[nt] < p.Cout) {
                            // first position attaining the max, scan order (0,0),(0,1),(1,0),(1,1) as torch's max_pool2d
                            const float m = fmaxf(v0, v1);
                            const int code = (a00 == m) ? 0 : (a01 == m) ? 1 : (a10 == m) ? 2 : 3;
                            p.pool_code[(((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nch[nt]] = (unsigned char)code;
                        }
                    }
        }
    }
#ifdef WITW_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP();
    if (lane == 0 && p.stamps != nullptr) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < n_stamp; ++i) o[i] = t_stamp[i];
        o[6] = __builtin_amdgcn_s_memrealtime() - r_stamp0;      // 100 MHz ticks over the same span as o[3] - o[0]
        o[7] = n_stamp;
    }
#endif
}

// One thread per packed float4: wpk[nt][kc][tap][q][n][0..3] <- w[cout][cin][kh][kw]
// (torch KCRS). transpose_flip!=0 builds the dgrad filter instead: roles of cin/cout
// swapped and taps rotated by 180 degrees (w'[ci][co][kh][kw] = w[co][ci][2-kh][2-kw]).
// taps = 9: all of the window; taps = 4: the 2x2 sub-window (tap_base + {0,1})^2 of the PACKED filter (TAPS = 4 kernels).
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin,
                                    int n_tiles, int nkc, int TN, int transpose_flip, int src_cout, int src_cin, int taps,
                                    int tap_base) {
    const size_t total = (size_t)n_tiles * nkc * taps * 2 * TN;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t t = idx;
    const int n = t % TN; t /= TN;
    const int q = t % 2; t /= 2;
    const int tap = t % taps; t /= taps;
    const int kc = t % nkc; t /= nkc;
    const int nt = (int)t;
    const int kh = (taps == 4) ? tap_base + (tap >> 1) : tap / 3, kw = (taps == 4) ? tap_base + (tap & 1) : tap % 3;
    const int co = nt * TN + n;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = kc * 8 + q * 4 + j;
        if (co < Cout && ci < Cin) {
            if (!transpose_flip)
                v[j] = w[(((size_t)co * src_cin + ci) * 3 + kh) * 3 + kw];
            else
                v[j] = w[(((size_t)ci * src_cin + co) * 3 + (2 - kh)) * 3 + (2 - kw)];
        }
    }
    reinterpret_cast<f32x4*>(wpk)[idx] = v;
}

// NCHW [B,C,H,W] -> NHWC8 [B,H,W,8], channels >= C zero filled (C <= 8).
__global__ void nchw_to_nhwc8_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int H, int W) {
    const size_t npix = (size_t)B * H * W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const size_t hw = (size_t)H * W;
    const size_t b = idx / hw, r = idx - b * hw;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (c < C) ? x[(b * C + c) * hw + r] : 0.f;
    f32x4* o = reinterpret_cast<f32x4*>(y + idx * 8);
    o[0] = f32x4{v[0], v[1], v[2], v[3]};
    o[1] = f32x4{v[4], v[5], v[6], v[7]};
}

// NCHW [B,C,H,W] -> NHWC [B,H,W,Cp] (Cp >= C, extra channels zero) and back; small tensors (embedding grads).
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int Cp, size_t hw, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % Cp;
    const size_t t = idx / Cp;
    const size_t b = t / hw, r = t - b * hw;
    y[idx] = (c < C) ? x[(b * C + c) * hw + r] : 0.f;
}

template <int TN, int SH, bool POOL, int NW, int GEO, int TAPS = 9>
int launch_conv_nw(ConvArgs a, hipStream_t st) {
    a.tiles_y = cdiv(a.Ho, GEO == 1 ? 4 * NW : NW);
    a.tiles_x = cdiv(a.Wo, GEO == 1 ? 16 : 64);
    const long long sp_total = (long long)a.B * a.tiles_x * a.tiles_y;
    a.n_tiles = cdiv(a.Cout, TN);
    a.sp_per_xcd = (int)((sp_total + 7) / 8);
    // fewer spatial tiles than XCDs: the XCD-aware order would put every workgroup (all n tiles and K slices of a spatial tile) on
    // sp_total of the 8 XCDs -- cvig_baseline's last block, 2 mosaics x 4 n tiles x 32 K slices, ran on 64 of the 256 CUs
    if (sp_total < 8) a.xcd_map = 0;
    const long long grid = a.xcd_map ? 8LL * a.sp_per_xcd * a.n_tiles : sp_total * a.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffLL || sp_total > 0x7fffffffLL) {
        witw_set_error("conv3x3: grid %lld out of range", grid);
        return WITW_ERR_INVALID;
    }
    a.sp_total = (int)sp_total;
    const unsigned gy = (TAPS == 4 && a.ksplit > 1) ? (unsigned)a.ksplit : 1u;
    hipLaunchKernelGGL((conv3x3_nhwc_f32_kernel<TN, SH, POOL, NW, GEO, TAPS>), dim3((unsigned)grid, gy), dim3(64 * NW), 0, st, a);
    WITW_CHECK_LAUNCH("conv3x3_nhwc_f32");
    witw_note_variant("conv3x3_nhwc_f32_kernel<%d,%d,%s,%d,%d,%d>", TN, SH, POOL ? "true" : "false", NW, GEO, TAPS);
    return WITW_OK;
}

// 8-wave workgroups (8-row tiles) when no rows are wasted and the grid fills its rounds of one workgroup per CU (api.hip)
int choose_waves(int B, int Ho, int Wo, int Cout, int force_nw) {
    const int TN = (Cout >= 128) ? 128 : 64;
    const long long big = (long long)cdiv(Cout, TN) * B * cdiv(Wo, TW_WIDE) * cdiv(Ho, 8);
    if (force_nw == 4) return 4;
    return (force_nw == 8 || ((Ho % 8) == 0 && witw_fills_rounds(big))) ? 8 : 4;
}

// narrow geometry (2x16-pixel M-tiles, 16-column workgroup tiles) for maps of at most 32 columns
bool choose_narrow(int Wo, int force_geo) { return force_geo == 1 || (force_geo != 0 && Wo <= 32); }

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);      // tuning / test aids: WITW_CONV_NW in {4,8}, WITW_CONV_GEO in {0,1}
    return e ? atoi(e) : dflt;
}

// 1 (default): the dgrad of a stride-(2,1) layer on the 8-wave 128-channel tile skips the products with zero-interleaved rows
// (GEO = 2); 0: it multiplies them like any other (WITW_CONV_DILSKIP=0 / witw_conv3x3_dil_skip(0): A/B runs, the bitwise test)
int g_dil_skip = -1;
int dil_skip() {
    if (g_dil_skip < 0) {
        const char* e = getenv("WITW_CONV_DILSKIP");
        g_dil_skip = e ? (atoi(e) != 0) : 1;
    }
    return g_dil_skip;
}

template <int TN, int SH, bool POOL>
int launch_conv(const ConvArgs& a, hipStream_t st) {
#ifndef WITW_NO_NARROW
    if (choose_narrow(a.Wo, a.force_geo)) return launch_conv_nw<TN, SH, POOL, 4, 1>(a, st);
#endif
    if (choose_waves(a.B, a.Ho, a.Wo, a.Cout, a.force_nw) == 8) {
        if constexpr (TN == 128 && SH == 1 && !POOL) {
            if (a.dil_h && dil_skip()) return launch_conv_nw<TN, SH, POOL, 8, 2>(a, st);      // dead (M-tile, tap) products not issued
        }
        return launch_conv_nw<TN, SH, POOL, 8, 0>(a, st);
    }
    return launch_conv_nw<TN, SH, POOL, 4, 0>(a, st);
}

// 2x2 sub-window kernels (stride 1, no pool)
template <int TN>
int launch_conv_taps4(const ConvArgs& a, hipStream_t st) {
#ifndef WITW_NO_NARROW
    if (choose_narrow(a.Wo, a.force_geo)) return launch_conv_nw<TN, 1, false, 4, 1, 4>(a, st);
#endif
    if (choose_waves(a.B, a.Ho, a.Wo, a.Cout, a.force_nw) == 8) return launch_conv_nw<TN, 1, false, 8, 0, 4>(a, st);
    return launch_conv_nw<TN, 1, false, 4, 0, 4>(a, st);
}

}  // namespace

#ifdef WITW_STAMPS
unsigned long long* witw_conv_stamps_ptr = nullptr;
#endif

extern "C" {

int witw_conv3x3_tile_n(int cout) { return cout >= 128 ? 128 : 64; }

// waves per workgroup (4 or 8) the launcher picks for a layer: names the kernel instantiation
// conv3x3_nhwc_f32_kernel<tile_n, stride_h, pool, waves> that a profile will show
int witw_conv3x3_workgroup_waves(int B, int H, int W, int Cout, int stride_h) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (stride_h != 1 && stride_h != 2)) return -1;
    if (choose_narrow(W, env_int("WITW_CONV_GEO", -1))) return 4;
    return choose_waves(B, (H + 2 - 3) / stride_h + 1, W, Cout, env_int("WITW_CONV_NW", 0));
}

long long witw_conv3x3_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return -1;
    const int TN = witw_conv3x3_tile_n(cout);
    const long long n_tiles = cdiv(cout, TN), nkc = cdiv(cin, 8);
    return n_tiles * nkc * 9 * 2 * TN * 4;
}

int witw_conv3x3_bias_floats(int cout) {
    const int TN = witw_conv3x3_tile_n(cout);
    return cdiv(cout, TN) * TN;
}

int witw_conv3x3_pack_weights(const float* w_kcrs, float* wpk, int cout, int cin, int transpose_flip, void* stream) {
    WITW_CHECK_ARG(w_kcrs && wpk, "pack_weights: null pointer");
    WITW_CHECK_ARG(cout > 0 && cin > 0, "pack_weights: bad shape cout=%d cin=%d", cout, cin);
    const int TN = witw_conv3x3_tile_n(cout);
    const int n_tiles = cdiv(cout, TN), nkc = cdiv(cin, 8);
    const size_t total = (size_t)n_tiles * nkc * 9 * 2 * TN;
    // in transpose_flip mode (cout,cin) describe the PACKED filter; the source tensor is [cin][cout][3][3]
    const int src_cin = transpose_flip ? cout : cin;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w_kcrs, wpk, cout, cin, n_tiles, nkc, TN, transpose_flip, cout, src_cin, 9, 0);
    WITW_CHECK_LAUNCH("pack_weights");
    return WITW_OK;
}

// ---- 2x2 sub-window form (cvig_baseline's Conv2d(k=4,s=2) over the space-to-depth image, model/cvig_baseline.py:236-252)
long long witw_conv3x3_packed_floats_taps4(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return -1;
    const int TN = witw_conv3x3_tile_n(cout);
    return (long long)cdiv(cout, TN) * cdiv(cin, 8) * 4 * 2 * TN * 4;
}

// Packs 4 of the 9 taps of a [cout][cin][3][3] filter: rows/columns {1,2} (transpose_flip = 0: the forward filter whose
// first tap row/column are zero) or, with transpose_flip, rows/columns {0,1} of the transposed, 180-degree rotated filter
// (its dgrad filter, source tensor [cin][cout][3][3] as in witw_conv3x3_pack_weights).
int witw_conv3x3_pack_weights_taps4(const float* w_kcrs, float* wpk, int cout, int cin, int transpose_flip, void* stream) {
    WITW_CHECK_ARG(w_kcrs && wpk, "pack_weights_taps4: null pointer");
    WITW_CHECK_ARG(cout > 0 && cin > 0, "pack_weights_taps4: bad shape cout=%d cin=%d", cout, cin);
    const int TN = witw_conv3x3_tile_n(cout);
    const int n_tiles = cdiv(cout, TN), nkc = cdiv(cin, 8);
    const size_t total = (size_t)n_tiles * nkc * 4 * 2 * TN;
    const int src_cin = transpose_flip ? cout : cin;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w_kcrs, wpk, cout, cin, n_tiles, nkc, TN, transpose_flip, cout, src_cin, 4, transpose_flip ? 0 : 1);
    WITW_CHECK_LAUNCH("pack_weights_taps4");
    return WITW_OK;
}

// workgroups the taps4 launcher would start for this shape (ksplit = 1)
static long long taps4_workgroups(int B, int H, int W, int Cout) {
    const int TN = witw_conv3x3_tile_n(Cout);
    const long long nt = cdiv(Cout, TN);
    if (choose_narrow(W, env_int("WITW_CONV_GEO", -1))) return nt * B * cdiv(H, 16) * cdiv(W, 16);
    const int nw = choose_waves(B, H, W, Cout, env_int("WITW_CONV_NW", 0));
    return nt * B * cdiv(H, nw) * cdiv(W, 64);
}

// Split-K factor the taps4 conv should run with: 1 when the plain launch already fills the chip (>= 2 workgroups per CU) or K
// is short; otherwise enough K slices for ~4 workgroups per CU, each at least 32 K-chunks (256 input channels) long. The deep
// cvig_baseline layers (16x16 .. 4x4 maps, K = 4 taps x 2048) start 8 .. 128 workgroups on 256 CUs without it.
int witw_conv3x3_taps4_ksplit(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
    const int forced = env_int("WITW_CONV_KSPLIT", 0);
    const int nkc = Cin >> 3;
    const long long wg = taps4_workgroups(B, H, W, Cout);
    int S = 1;
    if (forced > 0) S = forced;
    else if (wg < 512 && nkc >= 64) {
        S = (int)std::min<long long>(32, (1024 + wg - 1) / wg);
        // at least 32 chunks per slice -- 8 when the whole launch would not even reach one workgroup per CU otherwise (the 1 x 1 map of
        // cvig_baseline's last block: 8 tiles; 8 slices of 32 chunks ran 125 us on 64 CUs)
        S = std::min(S, wg * (nkc / 32) >= 256 ? nkc / 32 : nkc / 8);
    }
    S = std::max(1, std::min(S, nkc));
    const int per = cdiv(nkc, S);
    return cdiv(nkc, per);            // no empty slice
}

// y = act(conv over the 2x2 taps (tap_base + {0,1})^2 of the 3x3 window, zero padding 1) [* post_scale + post_shift],
// gate as in witw_conv3x3_fwd_ex. tap_base = 1 with a forward-packed filter, 0 with a transpose_flip-packed one.
//   ksplit > 1: split-K. y is a workspace of ksplit * B*H*W*Cout floats that receives the RAW partial sums of each K slice
//     ([ksplit][B,H,W,Cout]); bias / activation / affine are NOT applied (witw_taps4_splitk_finish does that), gate must be null.
//   s2d != 0 (ksplit == 1): y is the space-to-depth(2) image of the valid region, [B, ceil(valid_h/2), ceil(valid_w/2), 4*Cout]
//     with channel ((row&1)*2 + (col&1))*Cout + c and zeros outside the valid region: what the next k=4,s=2 layer reads.
int witw_conv3x3_fwd_taps4_ex(const float* x, const float* wpk4, const float* bias, const float* gate, const float* post_scale,
                              const float* post_shift, float* y, int B, int H, int W, int Cin, int Cout, int relu,
                              float lrelu_slope, int tap_base, int ksplit, int s2d, int valid_h, int valid_w, void* stream) {
    WITW_CHECK_ARG(x && wpk4 && bias && y, "conv3x3_fwd_taps4: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_fwd_taps4: bad shape B=%d H=%d W=%d Cout=%d", B, H, W, Cout);
    WITW_CHECK_ARG(Cin > 0 && (Cin % 8) == 0, "conv3x3_fwd_taps4: Cin=%d must be a positive multiple of 8", Cin);
    WITW_CHECK_ARG(relu >= 0 && relu <= 2, "conv3x3_fwd_taps4: activation %d unknown (0 none, 1 ReLU, 2 LeakyReLU)", relu);
    WITW_CHECK_ARG(tap_base == 0 || tap_base == 1, "conv3x3_fwd_taps4: tap_base=%d outside {0,1}", tap_base);
    WITW_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "conv3x3_fwd_taps4: post_scale and post_shift go together");
    WITW_CHECK_ARG(ksplit >= 1 && ksplit <= 65535 && ksplit <= (Cin >> 3), "conv3x3_fwd_taps4: ksplit=%d outside [1, Cin/8]", ksplit);
    WITW_CHECK_ARG(!(ksplit > 1 && (gate || s2d)), "conv3x3_fwd_taps4: split-K writes raw partial sums (no gate, no s2d output)");
    WITW_CHECK_ARG(!((ksplit > 1 || s2d) && (Cout & 3)), "conv3x3_fwd_taps4: split-K / s2d output need Cout %% 4 == 0, got %d", Cout);
    WITW_CHECK_ARG(!(s2d && gate), "conv3x3_fwd_taps4: s2d output with gate unsupported");
    WITW_CHECK_ARG(!s2d || (valid_h > 0 && valid_w > 0 && valid_h <= H && valid_w <= W),
                   "conv3x3_fwd_taps4: valid region %dx%d outside the %dx%d map", valid_h, valid_w, H, W);
    WITW_CHECK_ARG(!s2d || (2 * ((valid_h + 1) / 2) <= H && 2 * ((valid_w + 1) / 2) <= W),
                   "conv3x3_fwd_taps4: s2d output of an odd valid size needs one more computed row / column");
    ConvArgs a;
    a.x = x; a.wpk = wpk4; a.bias = bias; a.dropmask = nullptr; a.gate = gate; a.y = y;
    a.post_scale = post_scale; a.post_shift = post_shift; a.lrelu = lrelu_slope;
    a.pool_code = nullptr;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Ho = H; a.Wo = W;
    a.tiles_x = 0; a.tiles_y = 0;
    a.force_nw = env_int("WITW_CONV_NW", 0);
    a.force_geo = env_int("WITW_CONV_GEO", -1);
    a.xcd_map = env_int("WITW_CONV_XCD", 1) != 0;
    a.circ = 0; a.relu = relu; a.out_nchw = 0; a.dil_h = 0; a.tap_base = tap_base;
    a.ksplit = ksplit;
    a.kc_per_split = cdiv(Cin >> 3, ksplit);
    WITW_CHECK_ARG((long long)(ksplit - 1) * a.kc_per_split < (Cin >> 3), "conv3x3_fwd_taps4: ksplit=%d leaves an empty K slice", ksplit);
    a.split_stride = (size_t)B * H * W * Cout;
    a.s2d = s2d ? 1 : 0;
    a.valid_h = valid_h; a.valid_w = valid_w;
    a.s2d_h = (valid_h + 1) / 2; a.s2d_w = (valid_w + 1) / 2;
#ifdef WITW_STAMPS
    a.stamps = witw_conv_stamps_ptr;
#endif
    hipStream_t st = (hipStream_t)stream;
    return witw_conv3x3_tile_n(Cout) == 128 ? launch_conv_taps4<128>(a, st) : launch_conv_taps4<64>(a, st);
}

int witw_conv3x3_fwd_taps4(const float* x, const float* wpk4, const float* bias, const float* gate, const float* post_scale,
                           const float* post_shift, float* y, int B, int H, int W, int Cin, int Cout, int relu,
                           float lrelu_slope, int tap_base, void* stream) {
    return witw_conv3x3_fwd_taps4_ex(x, wpk4, bias, gate, post_scale, post_shift, y, B, H, W, Cin, Cout, relu, lrelu_slope,
                                     tap_base, 1, 0, 0, 0, stream);
}

int witw_nchw_to_nhwc8(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    WITW_CHECK_ARG(x && y, "nchw_to_nhwc8: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "nchw_to_nhwc8: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    const size_t npix = (size_t)B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       B, C, H, W);
    WITW_CHECK_LAUNCH("nchw_to_nhwc8");
    return WITW_OK;
}

int witw_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, int Cpad, void* stream) {
    WITW_CHECK_ARG(x && y, "nchw_to_nhwc: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_to_nhwc: bad shape");
    const size_t total = (size_t)B * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, C,
                       Cpad, (size_t)H * W, total);
    WITW_CHECK_LAUNCH("nchw_to_nhwc");
    return WITW_OK;
}

int witw_conv3x3_fwd_ex(const float* x, const float* wpk, const float* bias, const float* dropmask, const float* gate,
                        const float* post_scale, const float* post_shift, float* y, unsigned char* pool_code, int B, int H,
                        int W, int Cin, int Cout, int stride_h, int pad_circular, int relu, float lrelu_slope, int pool,
                        int out_nchw, int dilate_h, void* stream) {
    WITW_CHECK_ARG(x && wpk && bias && y, "conv3x3_fwd: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_fwd: bad shape B=%d H=%d W=%d Cout=%d", B, H, W, Cout);
    WITW_CHECK_ARG(Cin > 0 && (Cin % 8) == 0, "conv3x3_fwd: Cin=%d must be a positive multiple of 8", Cin);
    WITW_CHECK_ARG(stride_h == 1 || stride_h == 2, "conv3x3_fwd: stride_h=%d unsupported", stride_h);
    WITW_CHECK_ARG(!(pool && stride_h == 2), "conv3x3_fwd: pool with stride 2 unsupported");
    WITW_CHECK_ARG(!(pool && out_nchw), "conv3x3_fwd: pool with NCHW output unsupported");
    WITW_CHECK_ARG(!(pool && gate), "conv3x3_fwd: pool with gate unsupported");
    WITW_CHECK_ARG(!(dilate_h && stride_h == 2), "conv3x3_fwd: dilated input with stride 2 unsupported");
    WITW_CHECK_ARG(relu >= 0 && relu <= 2, "conv3x3_fwd: activation %d unknown (0 none, 1 ReLU, 2 LeakyReLU)", relu);
    WITW_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "conv3x3_fwd: post_scale and post_shift go together");
    WITW_CHECK_ARG(!(pool && relu == 2) && !(pool && post_scale), "conv3x3_fwd: pool needs a monotone epilogue (ReLU only)");
    ConvArgs a;
    a.x = x; a.wpk = wpk; a.bias = bias; a.dropmask = dropmask; a.gate = gate; a.y = y;
    a.post_scale = post_scale; a.post_shift = post_shift; a.lrelu = lrelu_slope;
    a.pool_code = pool ? pool_code : nullptr;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Ho = (H + 2 - 3) / stride_h + 1;
    a.Wo = W;
    a.tiles_x = 0;
    a.tiles_y = 0;   // set by the launcher for the chosen tile height
    a.force_nw = env_int("WITW_CONV_NW", 0);
    a.force_geo = env_int("WITW_CONV_GEO", -1);
    a.xcd_map = env_int("WITW_CONV_XCD", 1) != 0;      // 0: plain n-tile-major order (A/B timing)
    a.circ = pad_circular; a.relu = relu; a.out_nchw = out_nchw; a.dil_h = dilate_h; a.tap_base = 0;
    a.ksplit = 1; a.kc_per_split = 0; a.split_stride = 0; a.s2d = 0; a.s2d_h = a.s2d_w = a.valid_h = a.valid_w = 0;
#ifdef WITW_STAMPS
    a.stamps = witw_conv_stamps_ptr;
#endif
    hipStream_t st = (hipStream_t)stream;
    const int TN = witw_conv3x3_tile_n(Cout);
    if (TN == 128) {
        if (stride_h == 2) return launch_conv<128, 2, false>(a, st);
        return pool ? launch_conv<128, 1, true>(a, st) : launch_conv<128, 1, false>(a, st);
    }
    if (stride_h == 2) return launch_conv<64, 2, false>(a, st);
    return pool ? launch_conv<64, 1, true>(a, st) : launch_conv<64, 1, false>(a, st);
}

// 1 (default): a zero-interleaved launch (dilate_h) on the 8-wave 128-channel tile does not issue the MFMAs whose input rows are
// the interleaved zeros (half of them; same bits); 0: it issues them all. enable < 0 only queries. Returns the previous setting.
int witw_conv3x3_dil_skip(int enable) {
    const int prev = dil_skip();
    if (enable >= 0) g_dil_skip = enable != 0;
    return prev;
}

int witw_conv3x3_fwd(const float* x, const float* wpk, const float* bias, const float* dropmask, float* y, int B, int H,
                     int W, int Cin, int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw,
                     void* stream) {
    return witw_conv3x3_fwd_ex(x, wpk, bias, dropmask, nullptr, nullptr, nullptr, y, nullptr, B, H, W, Cin, Cout, stride_h,
                               pad_circular, relu ? 1 : 0, 0.f, pool, out_nchw, 0, stream);
}

}  // extern "C"
