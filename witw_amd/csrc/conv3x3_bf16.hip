// 3x3 convolution, NHWC bf16 operands / fp32 accumulate, implicit GEMM on v_mfma_f32_32x32x16_bf16 (gfx950).
//
// The bf16 inference path of the FOV_DSM encoder (BASELINE config "cvig_semantic.py semantic-branch variant,
// bf16 MFMA": reference layers model/cvig_semantic.py:275-325 = model/cvig_fov.py:256-294 with a 5-channel
// first conv). Same structure as the fp32 kernel (conv3x3.hip): NW x 64-pixel output tile x TN channels per
// workgroup, halo tile + weight slab double buffered in LDS as 16-byte slots in [group][row][col] /
// [tap][group][n] order, padding as a load policy (range-checked buffer loads), bias / ReLU / 2x2 max-pool in
// the epilogue. What changes with the data type:
//   * a 16-byte slot holds 8 bf16 channels; lanes 0-31 read channel group 0 and lanes 32-63 group 1 of a
//     16-channel K chunk, which is exactly one v_mfma_f32_32x32x16_bf16 operand pair (A[row][k=8h+j],
//     B[k=8h+j][col]) per ds_read_b128 — one MFMA (32 cycles) per fragment pair instead of four 64-cycle ones,
//     so the kernel is bound by operand delivery (LDS / L2), not by the matrix pipe;
//   * activations are stored bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32), channel count padded to 16;
//     the last layer writes the fp32 NCHW embedding.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 64;
constexpr int IW = TW + 2;

// Staging mode: 1 = input tile global -> VGPR -> ds_write, weight slab by LDS-DMA (buffer_load ... lds:
// the packed slab is already the LDS image, 1 KB contiguous per wave instruction); 2 = input tile by LDS-DMA too.
#ifndef WITW_BF_DMA
#define WITW_BF_DMA 1
#endif
#ifndef WITW_BF_SWAP
#define WITW_BF_SWAP 0          // A/B builds: 1 = MFMA operands swapped (lane = pixel, register quad = 4 consecutive channels), the
#endif                          //     epilogue transposes through LDS with 8-byte writes (12 instead of 40 LDS instructions per M-tile).
                                //     Parity-green, but measured SLOWER per tile (epilogue 9.1 k -> 11.9 k ticks)
#ifndef WITW_BF_S16_DMA
#define WITW_BF_S16_DMA 0       // 16x16x32 kernel: 1 = the input tile moves by LDS-DMA as well (A/B builds)
#endif
#ifndef WITW_BF_S16_PLANE16
#define WITW_BF_S16_PLANE16 1   // 16x16x32 kernel: channel-group planes of a stage at a pitch of 0 mod 16 slots (conflict-free A reads); 0 = round 2-5 layout
#endif
#ifndef WITW_BF_S16_SPREAD
#define WITW_BF_S16_SPREAD 3    // half-units (of 8 per chunk) over which the 16x16x32 kernel issues the staging pieces of the next chunk
#endif
#ifndef WITW_BF_SPREAD
#define WITW_BF_SPREAD 5        // taps over which the staging pieces of a chunk are issued (1 = all at tap 0)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor (base, stride 0, byte count, the flag word __builtin_amdgcn_make_buffer_rsrc is given elsewhere
// in this file) as four SGPRs for the hand-issued LDS-DMA loads.
__device__ __forceinline__ i32x4 raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// One wave instruction of LDS-DMA: lane l moves 16 B from rs[voff_l + soff] to LDS byte address lds_addr + 16*l
// (out-of-range lanes store zeros). Issued as inline assembly ON PURPOSE: for the builtin form the compiler puts a
// vmcnt wait in front of every later ds_read (it cannot tell the stage being read from the stage being filled), which
// serialises the pipeline; here the wave drains vmcnt itself once per K chunk (stage_wait) before the barrier.
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");      // m0 is a scratch register for the compiler too: it re-sets it right before each of its own uses
#endif
}

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

struct ConvBfArgs {
    const unsigned short* x;   // [B,H,W,Cin] NHWC bf16, Cin % 16 == 0
    const unsigned short* wpk; // packed bf16: [n_tile][cin/16][tap][group][TN][8]
    const float* bias;         // [n_tiles*TN] fp32 (zero padded)
    void* y;                   // NHWC bf16 [B,Hy,Wy,Cout] or fp32 NCHW [B,Cout,Hy,Wy]
    int B, H, W, Cin, Cout;
    int Ho, Wo;
    int tiles_x, tiles_y;
    int circ, relu, out_nchw_f32;
    int n_tiles, sp_total, sp_per_xcd, xcd_map;
    const float* dropmask;       // nullptr, or [B,Cout] Dropout2d scale applied to conv + bias before the ReLU
    const unsigned short* gate;  // nullptr, or a bf16 tensor shaped like y: outputs where gate <= 0 are zeroed (ReLU backward)
    int dil_h;                   // 1: input rows are zero-interleaved (logical row 2i = physical row i): dgrad of a stride-(2,1) conv
    unsigned char* pool_code;    // nullptr, or [B,Hy,Wy,Cout] arg-max position (dy*2+dx) of the fused 2x2 max pool (Cout % 8 == 0)
#ifdef WITW_BF_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/bf16_stamps.cpp): per wave {loop, vmcnt wait, barrier wait} ticks
#endif
};

template <int TN, int SH, bool POOL, int NW>
__global__ __launch_bounds__(64 * NW) void conv3x3_nhwc_bf16_kernel(ConvBfArgs p) {
    constexpr int TH = NW;
    constexpr int NTHREADS = 64 * NW;
    constexpr int IH = (TH - 1) * SH + 3;
    constexpr int IN_S = 2 * IH * IW;           // 16-B slots of one input stage (2 channel groups)
    constexpr int IN_P = (IN_S + 63) / 64 * 64;  // ... padded to whole 64-slot wave instructions (LDS-DMA granule)
    constexpr int W_S = 9 * 2 * TN;             // 16-B slots of one weight stage (a multiple of 64)
    constexpr int STAGE_S = IN_P + W_S;
    constexpr int NIN = (IN_S + NTHREADS - 1) / NTHREADS;
    constexpr int NIN_D = (IN_P / 64 + NW - 1) / NW;     // LDS-DMA wave instructions per wave and stage
    constexpr int NWT_D = (W_S / 64 + NW - 1) / NW;
    constexpr int WGM = (TN == 128) ? NW / 2 : NW;
    constexpr int WM = (2 * TH) / WGM;
    constexpr int WN = 2;
    constexpr unsigned OOR = 0x80000000u;
    static_assert(W_S % 64 == 0, "weight stage must be whole wave instructions");
    static_assert(STAGE_S * 16 >= (NW / 2) * 32 * 64 * 4, "a stage must hold the epilogue slabs of half the waves");

    // two separately declared stages: the compiler can then tell the fragment reads of one stage from the LDS-DMA
    // writes into the other (distinct objects -> no vmcnt wait in front of every ds_read); stageB's extra slot takes
    // the masked-off register-path stores
    __shared__ u32x4 stageA[STAGE_S];
    __shared__ u32x4 stageB[STAGE_S + 1];
    u32x4* const dummy_slot = stageB + STAGE_S;

#ifdef WITW_BF_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave) & (NW - 1);      // the mask tells the compiler the range
    const int l31 = lane & 31, hq = lane >> 5;

    // block -> (n tile, spatial tile): XCD-aware order, see conv3x3.hip (block i runs on XCD i % 8)
    int ntile, sp;
    if (p.xcd_map) {
        const int g = blockIdx.x >> 3;
        ntile = g % p.n_tiles;
        sp = (blockIdx.x & 7) * p.sp_per_xcd + g / p.n_tiles;
        if (sp >= p.sp_total) return;
    } else {
        ntile = blockIdx.x / p.sp_total;
        sp = blockIdx.x - ntile * p.sp_total;
    }
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int b = sp / tiles_img;
    sp -= b * tiles_img;
    const int ty = sp / p.tiles_x;
    const int tx = sp - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = ntile * TN;
    const int nkc = p.Cin >> 4;

    // ---- staging descriptors: per-image / per-weight-tile buffer resources, fixed per-thread byte offsets,
    // K-chunk advance in the scalar offset (32 B per pixel per chunk)
    const int Hp = p.dil_h ? (p.H - 1) / 2 + 1 : p.H;      // physical rows of the input
    const size_t img_elems = (size_t)Hp * p.W * p.Cin;
    __amdgpu_buffer_rsrc_t in_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * img_elems), 0, (unsigned)(img_elems * 2), 0x00020000);
    const i32x4 w_rd = raw_rsrc(reinterpret_cast<const u32x4*>(p.wpk) + (size_t)ntile * nkc * W_S, (unsigned)nkc * W_S * 16u);
#if WITW_BF_DMA >= 2
    const i32x4 in_rd = raw_rsrc(p.x + (size_t)b * img_elems, (unsigned)(img_elems * 2));
#endif
    // byte offset of halo-tile slot (pixel pix, channel group q) in the image, or OOR (loads return zero: padding)
    auto in_offset = [&](bool in_range, int pix, int q) -> unsigned {
        const int r = pix / IW, c = pix - r * IW;
        int gr = oy0 * SH - 1 + r;
        int gc = ox0 - 1 + c;
        bool ok = in_range && gr >= 0 && gr < p.H;
        if (p.dil_h) {
            ok = ok && !(gr & 1);
            gr >>= 1;
        }
        if (p.circ) {
            gc %= p.W;
            if (gc < 0) gc += p.W;
        } else {
            ok = ok && gc >= 0 && gc < p.W;
        }
        return ok ? (unsigned)((((size_t)gr * p.W + gc) * p.Cin + q * 8) * 2) : OOR;
    };
#if WITW_BF_DMA >= 2
    unsigned gin[NIN_D];        // DMA: slot s = instr * 64 + lane IS the LDS position [group][pixel]
#pragma unroll
    for (int i = 0; i < NIN_D; ++i) {
        const int s = (wave + NW * i) * 64 + lane;
        const int q = s / (IH * IW);
        gin[i] = in_offset(s < IN_S, s - q * (IH * IW), q);
    }
#else
    unsigned gin[NIN];          // registers: slot s -> pixel s/2, group s%2 (a pixel's 32 B load as one segment)
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int s = tid + i * NTHREADS;
        gin[i] = in_offset(s < IN_S, s >> 1, s & 1);
    }
    u32x4 rin[NIN];
#endif
    const unsigned lane16 = (unsigned)lane * 16u;

#ifdef WITW_DIAG_NOIN
    constexpr bool DIAG_IN = false;     // diagnostic builds: stage the input tile / the weight slab only once (wrong results)
#else
    constexpr bool DIAG_IN = true;
#endif
#ifdef WITW_DIAG_NOW
    constexpr bool DIAG_W = false;
#else
    constexpr bool DIAG_W = true;
#endif
    bool first_stage = true;
    // Staging of one K chunk is cut into PIECES wave instructions per wave (input pieces first: they come over the
    // fabric, the weight pieces are L2 hits). piece p of chunk kc -> LDS stage in_s:
#if WITW_BF_DMA >= 2
    constexpr int P_IN = NIN_D;
#else
    constexpr int P_IN = NIN;
#endif
    constexpr int PIECES = P_IN + NWT_D;
    auto stage_piece = [&](int kc, u32x4* in_s, int pc) {
        const unsigned in_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(in_s));
        if (pc < P_IN) {
            if (DIAG_IN || first_stage) {
#if WITW_BF_DMA >= 2
                const int j = wave_u + NW * pc;
                if (NIN_D * NW == IN_P / 64 || j < IN_P / 64) dma16(in_rd, in_lds + (unsigned)j * 1024u, gin[pc], (unsigned)kc * 32u);
#else
                rin[pc] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, gin[pc], (unsigned)kc * 32u, 0);
#endif
            }
        } else if (DIAG_W || first_stage) {
            const int j = wave_u + NW * (pc - P_IN);
            if (NWT_D * NW == W_S / 64 || j < W_S / 64)
                dma16(w_rd, in_lds + (unsigned)(IN_P + j * 64) * 16u, lane16, (unsigned)kc * W_S * 16u + (unsigned)j * 1024u);
        }
    };
    // register-staged input tile: VGPR -> LDS
    auto stage_commit = [&](u32x4* in_s) {
        (void)in_s;
#if WITW_BF_DMA < 2
        if (DIAG_IN || first_stage) {
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int s = tid + i * NTHREADS;
                u32x4* dst = (NIN * NTHREADS == IN_S || s < IN_S) ? in_s + (s & 1) * (IH * IW) + (s >> 1) : dummy_slot;
                *dst = rin[i];
            }
        }
#endif
    };
    // LDS-DMA data has landed once this wave's vector-memory counter drains (then the workgroup barrier publishes it)
    auto stage_wait = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), expcnt / lgkmcnt untouched
    };

    const int wm = (TN == 128) ? (wave >> 1) : wave;
    const int wn = (TN == 128) ? (wave & 1) : 0;
    int trow[WM], tcol[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        if (TN == 128) {
            trow[mt] = 2 * wm + (mt >> 1);
            tcol[mt] = 32 * (mt & 1);
        } else {
            trow[mt] = 2 * (wm >> 1) + mt;
            tcol[mt] = 32 * (wm & 1);
        }
    }
    int abase[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) abase[mt] = hq * (IH * IW) + trow[mt] * SH * IW + tcol[mt] + l31;
    const int wbase = hq * TN + wn * 64 + l31;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    u32x4 fa[2][WM], fb[2][WN];
    auto read_frags = [&](int set, const u32x4* in_s, const u32x4* w_s, int tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) fa[set][mt] = in_s[abase[mt] + kh * IW + kw];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) fb[set][nt] = w_s[tap * 2 * TN + wbase + nt * 32];
    };
    auto mfma_tap = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#if WITW_BF_SWAP
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[set][nt]),
                                                                     __builtin_bit_cast(bf16x8, fa[set][mt]), acc[mt][nt], 0, 0, 0);
#else
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][mt]),
                                                                     __builtin_bit_cast(bf16x8, fb[set][nt]), acc[mt][nt], 0, 0, 0);
#endif
    };

#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) stage_piece(0, stageA, pc);
    stage_commit(stageA);
    stage_wait();
    first_stage = false;
    __syncthreads();
    read_frags(0, stageA, stageA + IN_P, 0);

#ifdef WITW_BF_STAMPS
    unsigned long long t_vm = 0, t_bar = 0;
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    // one K chunk: 9 taps of MFMAs out of stage `in_s` while chunk kc+1 moves into stage `in_n`
    auto chunk = [&](const u32x4* in_s, u32x4* in_n, int kc) {
        const int kn = (kc + 1 < nkc) ? kc + 1 : kc;
        const u32x4* w_s = in_s + IN_P;
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) {
            read_frags((tap + 1) & 1, in_s, w_s, tap + 1);
#ifndef WITW_DIAG_NOSTAGE
            // the pieces of the next chunk leave over taps 0..SPREAD-1, a few per tap (each costs the wave issue time
            // that the partner wave's MFMAs cover), the rest of the chunk is landing time before the barrier
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc)
                if (pc * WITW_BF_SPREAD / PIECES == tap) stage_piece(kn, in_n, pc);
            if (tap == 6) stage_commit(in_n);
#endif
            mfma_tap(tap & 1);
            if (tap != 0 && tap != 6) {     // one fragment read per MFMA (tap 0 / the commit tap are left to the scheduler)
#pragma unroll
                for (int i = 0; i < WM + WN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < WM * WN - (WM + WN); ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
#ifdef WITW_BF_STAMPS
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
#ifndef WITW_DIAG_NOSTAGE
        stage_wait();
#endif
#ifdef WITW_BF_STAMPS
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
#ifndef WITW_DIAG_NOBARRIER
        __syncthreads();
#endif
#ifdef WITW_BF_STAMPS
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
        t_vm += ts1 - ts0;
        t_bar += ts2 - ts1;
#endif
        read_frags(1, in_n, in_n + IN_P, 0);
        mfma_tap(0);
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) fa[0][mt] = fa[1][mt];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) fb[0][nt] = fb[1][nt];
    };
    for (int kc = 0; kc < nkc; kc += 2) {
        chunk(stageA, stageB, kc);
        if (kc + 1 < nkc) chunk(stageB, stageA, kc + 1);
    }
#ifdef WITW_BF_STAMPS
    const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();      // the slabs below reuse the stages

#if WITW_BF_SWAP
    // ---- epilogue, swapped-operand form. acc[mt][nt][r] = output of PIXEL m = l31 of M-tile mt, CHANNEL
    // nt*32 + 8*(r>>2) + 4*hq + (r&3) of this wave's 64: a register quad holds 4 consecutive channels of one pixel, so
    // the LDS transposition writes 8 bytes of bf16 per quad (8 ds_write_b64 per M-tile instead of 32 scalar writes) into
    // a [pixel][64 channels] slab of 128-byte rows whose 16-byte chunks are XOR-swizzled with the pixel index (and the
    // two 8-byte halves of a chunk swapped on odd pixel octets): writes spread over all banks, the read-back is
    // 4 ds_read_b128 per M-tile (8 channels of a pixel per lane, halves swapped back at compile time) -> 16-byte stores.
    {
        const int cbase = n0 + wn * 64;                 // first channel of this wave
        const int m = l31;
        // the bias of this lane's 8 channel quads is fetched once, up front (32 registers; fetching it where a quad is
        // finished put a global-load latency in front of every LDS write); the Dropout2d scale, present on three layers of
        // a training forward only, is fetched per quad
        f32x4 bv4[WN][4];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bv4[nt][g] = *reinterpret_cast<const f32x4*>(p.bias + cbase + nt * 32 + 8 * g + 4 * hq);
        f32x4 bq, dq = {1.f, 1.f, 1.f, 1.f};
        auto load_quad = [&](int nt, int g) {
            bq = bv4[nt][g];
            if (p.dropmask != nullptr) {
                const int c = cbase + nt * 32 + 8 * g + 4 * hq;
#pragma unroll
                for (int j = 0; j < 4; ++j) dq[j] = (c + j < p.Cout) ? p.dropmask[(size_t)b * p.Cout + c + j] : 1.f;
            }
        };
        auto fin = [&](float v, int nt, int g, int j) {      // conv + bias -> Dropout2d scale -> ReLU, after load_quad(nt, g)
            (void)nt; (void)g;
            v = (v + bq[j]) * dq[j];
            if (p.relu) v = fmaxf(v, 0.f);
            return v;
        };
        auto gate_open = [](unsigned short g) { return (g & 0x7fffu) != 0 && !(g & 0x8000u); };   // bf16 value > 0
        const int Hy = POOL ? (p.Ho >> 1) : p.Ho;
        const int Wy = POOL ? (p.Wo >> 1) : p.Wo;
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        constexpr int SLABB = 32 * 128;
        char* const slab0 = reinterpret_cast<char*>((wave & 1) ? stageB : stageA) + (wave >> 1) * (2 * SLABB);
        const int prow = lane >> 3, oct = lane & 7;
        // byte offset of (pixel row q, channel chunk k = nt*4 + g, half hq) inside a slab
        auto wr_off = [&](int q, int k) { return q * 128 + (((k ^ (q & 7)) << 4) | ((hq ^ ((q >> 3) & 1)) << 3)); };
        // read-back of one slab row group: pixel rows gq*8 + prow, chunk `oct`; halves come back swapped on odd gq
        auto rd_chunk = [&](const char* slab, int gq) {
            const u16x8 v = *reinterpret_cast<const u16x8*>(slab + (gq * 8 + prow) * 128 + ((oct ^ prow) << 4));
            return (gq & 1) ? (u16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]} : v;
        };
        auto store8 = [&](u16x8 o, int yy, int xx) {
            const int nbase = cbase + oct * 8;
            if (yy < Hy && xx < Wy && nbase < p.Cout) {
                const size_t off = (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nbase;
                if (p.gate != nullptr) {
                    const u16x8 gt = *reinterpret_cast<const u16x8*>(p.gate + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (!gate_open(gt[e])) o[e] = 0;
                }
                __builtin_nontemporal_store(o, reinterpret_cast<u16x8*>(reinterpret_cast<unsigned short*>(p.y) + off));
            }
        };
        if (p.out_nchw_f32 || (p.Cout & 7) != 0) {
            // fp32 NCHW embedding (last layer) or a ragged channel count: lanes are consecutive pixels of a row, so
            // every register is a coalesced run along x
            if (!POOL) {
#pragma unroll
                for (int mt = 0; mt < WM; ++mt) {
                    const int yy = oy0 + trow[mt], xx = ox0 + tcol[mt] + m;
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int g = r >> 2, j = r & 3;
                            const int c = cbase + nt * 32 + 8 * g + 4 * hq + j;
                            if (j == 0) load_quad(nt, g);
                            float v = fin(acc[mt][nt][r], nt, g, j);
                            if (yy < Hy && xx < Wy && c < p.Cout) {
                                if (p.out_nchw_f32) {
                                    reinterpret_cast<float*>(p.y)[(((size_t)b * p.Cout + c) * Hy + yy) * Wy + xx] = v;
                                } else {
                                    const size_t o = (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + c;
                                    if (p.gate != nullptr && !gate_open(p.gate[o])) v = 0.f;
                                    reinterpret_cast<__bf16*>(p.y)[o] = (__bf16)v;
                                }
                            }
                        }
                }
            } else {
#pragma unroll
                for (int pr = 0; pr < WM / 2; ++pr) {
                    const int mtA = (TN == 128) ? (pr & 1) : 0;
                    const int mtB = (TN == 128) ? (2 + (pr & 1)) : 1;
                    const int yy = (oy0 + trow[mtA]) >> 1, xx = ((ox0 + tcol[mtA]) >> 1) + (m >> 1);
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int g = r >> 2, j = r & 3;
                            const int c = cbase + nt * 32 + 8 * g + 4 * hq + j;
                            if (j == 0) load_quad(nt, g);
                            float v = fmaxf(acc[mtA][nt][r], acc[mtB][nt][r]);
                            v = fmaxf(v, __shfl_xor(v, 1, 64));
                            v = fin(v, nt, g, j);
                            if (!(m & 1) && yy < Hy && xx < Wy && c < p.Cout)
                                reinterpret_cast<__bf16*>(p.y)[(((size_t)b * Hy + yy) * Wy + xx) * p.Cout + c] = (__bf16)v;
                        }
                }
            }
        } else if (!POOL) {
            auto write_tile = [&](int mt, char* slab) {
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 q;
                        load_quad(nt, g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) q[j] = (__bf16)fin(acc[mt][nt][4 * g + j], nt, g, j);
                        *reinterpret_cast<bf16x4*>(slab + wr_off(m, nt * 4 + g)) = q;
                    }
            };
            write_tile(0, slab0);
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) {
                if (mt + 1 < WM) write_tile(mt + 1, slab0 + ((mt + 1) & 1) * SLABB);
                const char* slab = slab0 + (mt & 1) * SLABB;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    store8(rd_chunk(slab, gq), oy0 + trow[mt], ox0 + tcol[mt] + gq * 8 + prow);
            }
        } else {
            // fused 2x2 max-pool: vertical partner = the M-tile one row below (same lane), horizontal partner = the
            // neighbouring lane; even lanes keep the pooled pixel m/2 (16 per M-tile pair)
#pragma unroll
            for (int pr = 0; pr < WM / 2; ++pr) {
                const int mtA = (TN == 128) ? (pr & 1) : 0;
                const int mtB = (TN == 128) ? (2 + (pr & 1)) : 1;
                const int yy = (oy0 + trow[mtA]) >> 1;
                const int xb = (ox0 + tcol[mtA]) >> 1;
                const int pc = m >> 1;
                char* slab = slab0 + (pr & 1) * SLABB;
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 q;
                        unsigned code4 = 0;
                        load_quad(nt, g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a00 = acc[mtA][nt][4 * g + j], a10 = acc[mtB][nt][4 * g + j];
                            const float a01 = __shfl_xor(a00, 1, 64), a11 = __shfl_xor(a10, 1, 64);
                            const float mx = fmaxf(fmaxf(a00, a01), fmaxf(a10, a11));
                            q[j] = (__bf16)fin(mx, nt, g, j);
                            // first position attaining the max, scan order (0,0),(0,1),(1,0),(1,1) as torch's max_pool2d
                            const unsigned code = (a00 == mx) ? 0u : (a01 == mx) ? 1u : (a10 == mx) ? 2u : 3u;
                            code4 |= code << (8 * j);
                        }
                        if (!(m & 1)) {
                            *reinterpret_cast<bf16x4*>(slab + wr_off(pc, nt * 4 + g)) = q;
                            const int c = cbase + nt * 32 + 8 * g + 4 * hq;
                            if (p.pool_code != nullptr && yy < Hy && xb + pc < Wy && c < p.Cout)
                                *reinterpret_cast<unsigned*>(p.pool_code + (((size_t)b * Hy + yy) * Wy + xb + pc) * p.Cout + c) = code4;
                        }
                    }
#pragma unroll
                for (int gq = 0; gq < 2; ++gq)
                    store8(rd_chunk(slab, gq), yy, xb + gq * 8 + prow);
            }
        }
    }
#else
    // ---- epilogue
    float bv[WN], dm[WN];
    int nch[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        nch[nt] = n0 + wn * 64 + nt * 32 + l31;
        bv[nt] = p.bias[nch[nt]];
        dm[nt] = 1.f;
        if (p.dropmask != nullptr && nch[nt] < p.Cout) dm[nt] = p.dropmask[(size_t)b * p.Cout + nch[nt]];
    }
    // conv + bias -> Dropout2d scale -> ReLU (the order of the fp32 kernel, model/cvig_fov.py:287-288)
    auto fin = [&](float v, int nt) {
        v = (v + bv[nt]) * dm[nt];
        if (p.relu) v = fmaxf(v, 0.f);
        return v;
    };
    auto gate_open = [](unsigned short g) { return (g & 0x7fffu) != 0 && !(g & 0x8000u); };   // bf16 value > 0
    const int Hy = POOL ? (p.Ho >> 1) : p.Ho;
    const int Wy = POOL ? (p.Wo >> 1) : p.Wo;
    auto emit = [&](float v, int nt, int yy, int xx) {
        v = fin(v, nt);
        if (yy < Hy && xx < Wy && nch[nt] < p.Cout) {
            if (p.out_nchw_f32) {
                reinterpret_cast<float*>(p.y)[(((size_t)b * p.Cout + nch[nt]) * Hy + yy) * Wy + xx] = v;
            } else {
                const size_t o = (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nch[nt];
                if (p.gate != nullptr && !gate_open(p.gate[o])) v = 0.f;
                reinterpret_cast<__bf16*>(p.y)[o] = (__bf16)v;
            }
        }
    };

    if (!POOL && !p.out_nchw_f32 && (p.Cout & 7) == 0) {
        // wide store: fp32 tile -> wave-private LDS slab -> 8 channels (16 B of bf16) per lane
        float* slab = reinterpret_cast<float*>((wave & 1) ? stageB : stageA) + (wave >> 1) * (32 * 64);
        const int prow = lane >> 3, pc8 = (lane & 7) * 8;   // read-back role: pixel row in a group of 8, channel octet
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[((r & 3) + 8 * (r >> 2) + 4 * hq) * 64 + nt * 32 + l31] = fin(acc[mt][nt][r], nt);
            const int yy = oy0 + trow[mt];
            const int nbase = n0 + wn * 64 + pc8;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = g * 8 + prow;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + m * 64 + pc8 + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (__bf16)v0[e];
                    o[4 + e] = (__bf16)v1[e];
                }
                const int xx = ox0 + tcol[mt] + m;
                if (yy < Hy && xx < Wy && nbase < p.Cout) {
                    const size_t off = (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nbase;
                    if (p.gate != nullptr) {
                        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                        const u16x8 gt = *reinterpret_cast<const u16x8*>(p.gate + off);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (!gate_open(gt[e])) o[e] = (__bf16)0.f;
                    }
                    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + off));
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
                    emit(acc[mt][nt][r], nt, oy0 + trow[mt], ox0 + tcol[mt] + m);
                }
    } else if ((p.Cout & 7) == 0) {
        // fused 2x2 max-pool, wide store through the wave-private slab: 16 pooled pixels x 64 channels per
        // M-tile pair leave as 16-byte stores of 8 bf16 channels
        float* slab = reinterpret_cast<float*>((wave & 1) ? stageB : stageA) + (wave >> 1) * (32 * 64);
        const int prow = lane >> 3, pc8 = (lane & 7) * 8;
#pragma unroll
        for (int pr = 0; pr < WM / 2; ++pr) {
            const int mtA = (TN == 128) ? (pr & 1) : 0;
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
            const int nbase = n0 + wn * 64 + pc8;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int pc = g * 8 + prow;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + pc * 64 + pc8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + pc * 64 + pc8 + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (__bf16)v0[e];
                    o[4 + e] = (__bf16)v1[e];
                }
                if (yy < Hy && xb + pc < Wy && nbase < p.Cout)
                    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + (((size_t)b * Hy + yy) * Wy + xb + pc) * p.Cout + nbase));
            }
        }
    } else {
#pragma unroll
        for (int pr = 0; pr < WM / 2; ++pr) {
            const int mtA = (TN == 128) ? (pr & 1) : 0;
            const int mtB = (TN == 128) ? (2 + (pr & 1)) : 1;
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v0 = fmaxf(acc[mtA][nt][4 * g + 2 * e], acc[mtA][nt][4 * g + 2 * e + 1]);
                        const float v1 = fmaxf(acc[mtB][nt][4 * g + 2 * e], acc[mtB][nt][4 * g + 2 * e + 1]);
                        const int yy = (oy0 + trow[mtA]) >> 1;
                        const int xx = ((ox0 + tcol[mtA]) >> 1) + 4 * g + 2 * hq + e;
                        emit(fmaxf(v0, v1), nt, yy, xx);
                    }
        }
    }
#endif      // WITW_BF_SWAP
#ifdef WITW_BF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && p.stamps != nullptr) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = t_loop1 - t_loop0;
        o[1] = t_vm;
        o[2] = t_bar;
        o[3] = 1;
        o[4] = t_loop0 - t_start;
        o[5] = __builtin_amdgcn_s_memtime() - t_loop1;
        o[6] = __builtin_amdgcn_s_memtime() - t_start;
        o[7] = __builtin_amdgcn_s_memrealtime() - r_start;     // 100 MHz ticks
    }
#endif
}

// Hand-issued fragment reads of the 16x16x32 kernel below: asm volatile keeps them where they are written (left to the scheduler,
// refills are hoisted above the MFMAs that still read the old fragment and cost a second register set: 49 spills), the matching
// s_waitcnt counts the reads issued after the one that is needed (LDS returns in order; extra LDS operations of the compiler
// in the queue only make a wait stricter).
__device__ __forceinline__ u32x4 s16_read(unsigned addr, int off) {      // off: a constant once the caller's loops are unrolled
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off));
    return v;
}
__device__ __forceinline__ void s16_wait(int n, u32x4& a) {
    switch (n) {      // n is a constant once the caller's loops are unrolled
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a)); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a)); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a)); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a)); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a)); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a)); break;
    default: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a)); break;
    }
}
__device__ __forceinline__ void s16_wait5(int n, u32x4& a, u32x4& b0, u32x4& b1, u32x4& b2, u32x4& b3) {
    switch (n) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)); break;
    default: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same convolution on v_mfma_f32_16x16x32_bf16 (inference forward of the 128-channel tile, 8 waves, stride 1, NHWC bf16 out,
// no gate / Dropout2d scale / pool codes: everything else runs on the kernel above). Same workgroup tile, same LDS stages and
// staging as above; what changes is the MFMA shape: the chip holds a higher clock on the 16x16x32 form (MI355X_MICROARCH.md,
// 'DVFS give-back' item 7; a probe build of the kernel above with equal FLOPs and LDS reads ran the whole step +7 %).
//   K = 32 of one MFMA = TWO taps of the 16-channel chunk: lane l holds 8 channels of pixel (l & 15) for k-group l >> 4 =
//   (tap of the pair, channel group). A fragment = 16 pixels of one row x {tap a, tap b} x 2 groups = one ds_read_b128 whose
//   lane address is base + (tap offset by lane), B fragment = 16 output channels likewise. A wave's 2 rows x 64 pixels x 64
//   channels are 8 x 4 tiles of 16 x 16 (4 accumulation registers each: register r of lane l = pixel 4*(l>>4) + r, channel l & 15).
//   The taps 0..7 of a chunk make 4 pairs; tap 8 of an even chunk shares its MFMA with tap 8 of the odd chunk behind it (k-groups
//   0,1 read stage A, k-groups 2,3 stage B), so no MFMA runs half empty: two chunks = 18 half-units of 16 MFMAs.
// TRAIN = the forms a training step needs on top of the plain forward: Dropout2d scale before the ReLU (dropmask), the ReLU gate of
// a dgrad launch (gate), zero-interleaved input rows (dil_h: dgrad of a stride-(2,1) layer), arg-max codes of the fused max-pool
// (pool_code). A template flag, so that the inference instantiations keep their register allocation (256 VGPRs, at the limit).
template <bool POOL, bool TRAIN>
__global__ __launch_bounds__(512) void conv3x3_bf16_s16_kernel(ConvBfArgs p) {
    constexpr int TN = 128, NW = 8, TH = 8, NTHREADS = 512;
    constexpr int IH = TH + 2;
    // Plane pitch of the two 8-channel groups of a stage. A fragment read (ds_read_b128) is served in four groups of 16 lanes
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32); in this kernel lanes 16-31 / 48-63 of a wave read channel group 1, so a
    // 16-lane group mixes pixels l15 of plane 0 with pixels l15' of plane 1 and is conflict-free only if the plane pitch is a
    // multiple of 16 slots (64 banks): IH*IW = 660 = 4 mod 16 put pixels 12-15 of plane 0 on the banks of pixels 8-11 of plane 1 --
    // every A read 2-way, 36-38 % of the LDS-active cycles (profiles/r05_bf16_train_lds_pmc.txt). 672 slots per plane: the stage
    // keeps its size (2 * 672 = 1344 = the 64-slot round-up of 1320 it already had).
    constexpr int PL = WITW_BF_S16_PLANE16 ? (IH * IW + 15) / 16 * 16 : IH * IW;
    constexpr int IN_S = 2 * PL;
    constexpr int IN_P = (IN_S + 63) / 64 * 64;
    constexpr int W_S = 9 * 2 * TN;
    constexpr int STAGE_S = IN_P + W_S;
    constexpr int NIN = (IN_S + NTHREADS - 1) / NTHREADS;
    constexpr int NWT_D = (W_S / 64 + NW - 1) / NW;
    constexpr unsigned OOR = 0x80000000u;
    constexpr int SLAB_P = 68;      // floats per slab row: 64 channels + 4 (the four 16-lane groups of a write land on different banks)
    static_assert(STAGE_S * 16 >= (NW / 2) * 32 * SLAB_P * 4, "a stage must hold the epilogue slabs of half the waves");

    __shared__ u32x4 stageA[STAGE_S];
    __shared__ u32x4 stageB[STAGE_S + 1];
    u32x4* const dummy_slot = stageB + STAGE_S;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave) & (NW - 1);
    const int l15 = lane & 15, kg = lane >> 4;
    const int sel = kg >> 1, grp = kg & 1;       // which tap of a pair / which 8-channel group this lane's k values are

    int ntile, sp;
    if (p.xcd_map) {
        const int g = blockIdx.x >> 3;
        ntile = g % p.n_tiles;
        sp = (blockIdx.x & 7) * p.sp_per_xcd + g / p.n_tiles;
        if (sp >= p.sp_total) return;
    } else {
        ntile = blockIdx.x / p.sp_total;
        sp = blockIdx.x - ntile * p.sp_total;
    }
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int b = sp / tiles_img;
    sp -= b * tiles_img;
    const int ty = sp / p.tiles_x;
    const int tx = sp - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = ntile * TN;
    const int nkc = p.Cin >> 4;

    // ---- staging: as in the kernel above (input tile through registers, weight slab by LDS-DMA)
    const int Hp = (TRAIN && p.dil_h) ? (p.H - 1) / 2 + 1 : p.H;      // physical rows of the input
    const size_t img_elems = (size_t)Hp * p.W * p.Cin;
    __amdgpu_buffer_rsrc_t in_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * img_elems), 0, (unsigned)(img_elems * 2), 0x00020000);
    const i32x4 w_rd = raw_rsrc(reinterpret_cast<const u32x4*>(p.wpk) + (size_t)ntile * nkc * W_S, (unsigned)nkc * W_S * 16u);
#if WITW_BF_S16_DMA
    // input tile by LDS-DMA too: slot s = instruction * 64 + lane IS the LDS position [group][pixel]; no register transit
    constexpr int NIN_D = (IN_P / 64 + NW - 1) / NW;
    const i32x4 in_rd = raw_rsrc(p.x + (size_t)b * img_elems, (unsigned)(img_elems * 2));
    unsigned gin[NIN_D];
#pragma unroll
    for (int i = 0; i < NIN_D; ++i) {
        const int s = (wave + NW * i) * 64 + lane;
        const int q = s / PL;
        const int pix = s - q * PL;
        const int r = pix / IW, c = pix - r * IW;
        int gr = oy0 - 1 + r;
        int gc = ox0 - 1 + c;
        bool ok = s < IN_S && pix < IH * IW && gr >= 0 && gr < p.H;
        if (TRAIN && p.dil_h) {
            ok = ok && !(gr & 1);
            gr >>= 1;
        }
        if (p.circ) {
            gc %= p.W;
            if (gc < 0) gc += p.W;
        } else {
            ok = ok && gc >= 0 && gc < p.W;
        }
        gin[i] = ok ? (unsigned)((((size_t)gr * p.W + gc) * p.Cin + q * 8) * 2) : OOR;
    }
    constexpr int P_IN = NIN_D;
#else
    constexpr int P_IN = NIN;
    unsigned gin[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int s = tid + i * NTHREADS;
        const int pix = s >> 1, q = s & 1;
        const int r = pix / IW, c = pix - r * IW;
        int gr = oy0 - 1 + r;
        int gc = ox0 - 1 + c;
        bool ok = s < IN_S && pix < IH * IW && gr >= 0 && gr < p.H;
        if (TRAIN && p.dil_h) {
            ok = ok && !(gr & 1);
            gr >>= 1;
        }
        if (p.circ) {
            gc %= p.W;
            if (gc < 0) gc += p.W;
        } else {
            ok = ok && gc >= 0 && gc < p.W;
        }
        gin[i] = ok ? (unsigned)((((size_t)gr * p.W + gc) * p.Cin + q * 8) * 2) : OOR;
    }
    u32x4 rin[NIN];
#endif
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr int PIECES = P_IN + NWT_D;
    auto stage_piece = [&](int kc, u32x4* in_s, int pc) {
        if (pc < P_IN) {
#if WITW_BF_S16_DMA
            const unsigned in_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(in_s));
            const int j = wave_u + NW * pc;
            if (NIN_D * NW == IN_P / 64 || j < IN_P / 64) dma16(in_rd, in_lds + (unsigned)j * 1024u, gin[pc], (unsigned)kc * 32u);
#else
            rin[pc] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, gin[pc], (unsigned)kc * 32u, 0);
#endif
        } else {
            const unsigned in_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(in_s));
            const int j = wave_u + NW * (pc - P_IN);
            if (NWT_D * NW == W_S / 64 || j < W_S / 64)
                dma16(w_rd, in_lds + (unsigned)(IN_P + j * 64) * 16u, lane16, (unsigned)kc * W_S * 16u + (unsigned)j * 1024u);
        }
    };
    auto stage_commit = [&](u32x4* in_s) {
#if WITW_BF_S16_DMA
        (void)in_s; (void)dummy_slot;
#else
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int s = tid + i * NTHREADS;
            u32x4* dst = (NIN * NTHREADS == IN_S || s < IN_S) ? in_s + (s & 1) * PL + (s >> 1) : dummy_slot;
            *dst = rin[i];
        }
#endif
    };
    auto stage_wait = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };      // vmcnt(0)

    // ---- this wave's tiles: rows 2*wm, 2*wm + 1 of the workgroup's 8, all 64 columns; channels wn*64 .. +63
    const int wm = wave >> 1, wn = wave & 1;
    // pixel tile a (0..7): row 2*wm + (a >> 2), columns 16*(a & 3) ..+15. A-fragment slot of this lane for tap offset `off`:
    //   grp*PL + (2*wm + (a>>2))*IW + 16*(a&3) + l15 + off  =  a_lane + off + [(a>>2)*IW + 16*(a&3): immediate]
    const int a_lane = grp * PL + 2 * wm * IW + l15;
    // channel tile bt (0..3): B-fragment slot = IN_P + tap*2*TN + grp*TN + wn*64 + 16*bt + l15 = w_lane + tap*2*TN + [16*bt]
    const int w_lane = IN_P + grp * TN + wn * 64 + l15;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) acc[a][bt] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 fa[4], fb[2][4];      // A: one set, refilled in place as soon as a tile's MFMAs are issued; B: one set per tap pair
    auto tap_off = [](int t) { return (t / 3) * IW + t % 3; };
    // Tap pairs are chosen so that the two taps of a pair differ by the same amount in most pairs and the lane-dependent part of
    // an address is one of few registers: P0 = (0,1), P1 = (3,4), P2 = (6,7) (one column apart), P3 = (2,5) (one row apart);
    // tap 8 goes into the shared pair.
    // lane part of an address + stage base, per stage (index 0: stage A, 1: stage B): ten registers, every other part of an
    // address is the immediate offset of the ds_read_b128
    const unsigned ldsA = lds_address(stageA), ldsB = lds_address(stageB);
    unsigned a_l1[2], a_lW[2], w_l1[2], w_l3[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const unsigned base = st ? ldsB : ldsA;
        a_l1[st] = base + (unsigned)(a_lane + sel) * 16u;
        a_lW[st] = base + (unsigned)(a_lane + sel * IW) * 16u;
        w_l1[st] = base + (unsigned)(w_lane + sel * 2 * TN) * 16u;
        w_l3[st] = base + (unsigned)(w_lane + sel * 3 * 2 * TN) * 16u;
    }
    // the shared pair [tap 8 of the even chunk (stage A) | tap 8 of the odd chunk (stage B)]: the stage is chosen by the lane
    const unsigned a_lS = (sel ? ldsB : ldsA) + (unsigned)(a_lane + tap_off(8)) * 16u;
    const unsigned w_lS = (sel ? ldsB : ldsA) + (unsigned)(w_lane + 8 * 2 * TN) * 16u;
    // A fragment i of half-unit (pr, h) (tiles 4h..4h+3) out of stage st
    auto read_a = [&](int st, int pr, int h, int i) -> u32x4 {      // st: 0 / 1 = a stage, 2 = the shared pair
        const int a = 4 * h + i;
        const int tile = (a >> 2) * IW + 16 * (a & 3);
        if (st == 2) return s16_read(a_lS, tile * 16);
        const int t0 = (pr == 0) ? 0 : (pr == 1) ? 3 : (pr == 2) ? 6 : 2;
        return s16_read((pr == 3) ? a_lW[st] : a_l1[st], (tap_off(t0) + tile) * 16);
    };
    auto read_b = [&](int st, int pr, int bt) -> u32x4 {
        if (st == 2) return s16_read(w_lS, 16 * bt * 16);
        const int t0 = (pr == 0) ? 0 : (pr == 1) ? 3 : (pr == 2) ? 6 : 2;
        return s16_read((pr == 3) ? w_l3[st] : w_l1[st], (t0 * 2 * TN + 16 * bt) * 16);
    };

#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) stage_piece(0, stageA, pc);
    stage_commit(stageA);
    stage_wait();
    __syncthreads();
    // the reads of a second-half unit leave in the order A0 B0 B1 | A1 B2 B3 | A2 | A3; the prologue is such a unit
    fa[0] = read_a(0, 0, 0, 0);
    fb[0][0] = read_b(0, 0, 0);
    fb[0][1] = read_b(0, 0, 1);
    fa[1] = read_a(0, 0, 0, 1);
    fb[0][2] = read_b(0, 0, 2);
    fb[0][3] = read_b(0, 0, 3);
    fa[2] = read_a(0, 0, 0, 2);
    fa[3] = read_a(0, 0, 0, 3);

    // Two K chunks (an even one in stage A, the odd one behind it in stage B) = 18 half-units of 16 MFMAs (4 pixel tiles x 4
    // channel tiles): units 0-7 the four tap pairs of the even chunk, 8-9 the SHARED pair [tap 8 of the even chunk | tap 8 of the
    // odd chunk] (no half-empty MFMA for the ninth tap), 10-17 the four pairs of the odd chunk. A tile's fragment is refilled for
    // the next unit right behind its four MFMAs (12 MFMAs of lead), the next pair's B fragments are read behind the first two
    // tiles of a pair's second half. Staging: the odd chunk leaves for stage B in units 0-4 (register part committed in unit 5),
    // the next even chunk for stage A in units 10-14 (15). Barriers: in front of unit 7 (its refills read the shared pair: stage B
    // must have landed), behind unit 9 (every wave has read tap 8 out of stage A before anyone overwrites it), in front of unit
    // 17 (its refills read the next even chunk). q = B set of this pair of chunks' first tap pair (9 pairs: it flips every time).
    auto chunk_pair = [&](int q, int kc) {
        const int kn2 = (kc + 2 < nkc) ? kc + 2 : kc;      // behind the last pair the even stage is refilled with itself (never read)
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            // what unit u multiplies
            const int h = u & 1;
            const int pseq = u >> 1;                       // pair number in the sequence: 0-3 even chunk, 4 shared, 5-8 odd chunk
            const int sb = (q + pseq) & 1;
            // what unit u + 1 reads (u = 17: unit 0 of the next chunk pair, out of stage A)
            const int un = (u + 1) % 18;
            const int hn = un & 1, pseqn = un >> 1;
            const int stn = (pseqn < 4) ? 0 : (pseqn == 4) ? 2 : 1;
            const int prn = (pseqn < 4) ? pseqn : (pseqn == 4) ? 4 : pseqn - 5;
            const int sbn = (u == 17) ? ((q + 9) & 1) : ((q + pseqn) & 1);
            if (u == 7 || u == 17) {
                stage_wait();
                __syncthreads();
            }
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc) {
                if (pc * WITW_BF_S16_SPREAD / PIECES == u) stage_piece(kc + 1, stageB, pc);
                if (pc * WITW_BF_S16_SPREAD / PIECES + 10 == u) stage_piece(kn2, stageA, pc);
            }
            if (u == 5) stage_commit(stageB);
            if (u == 15) stage_commit(stageA);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (h == 0) {       // the previous unit issued A0 B0 B1 A1 B2 B3 A2 A3
                    if (i == 0) s16_wait5(2, fa[0], fb[sb][0], fb[sb][1], fb[sb][2], fb[sb][3]);
                    else s16_wait(i == 1 ? 5 : 3, fa[i]);
                } else {            // the previous unit issued A0 A1 A2 A3
                    s16_wait(i == 0 ? 3 : i == 1 ? 5 : 7, fa[i]);
                }
#pragma unroll
                for (int bt = 0; bt < 4; ++bt)
                    acc[4 * h + i][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                                  __builtin_bit_cast(bf16x8, fb[sb][bt]), acc[4 * h + i][bt], 0, 0, 0);
                fa[i] = read_a(stn, prn, hn, i);
                if (h == 1 && i < 2) {
                    fb[sbn][2 * i] = read_b(stn, prn, 2 * i);
                    fb[sbn][2 * i + 1] = read_b(stn, prn, 2 * i + 1);
                }
            }
            if (u == 9) __syncthreads();
        }
    };
    for (int kc = 0; kc < nkc; kc += 4) {
        chunk_pair(0, kc);
        if (kc + 2 < nkc) chunk_pair(1, kc + 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the last unit's refills (never used) must have landed before the slabs
    __syncthreads();      // the slabs below reuse the stages

    // ---- epilogue: bias + ReLU (+ 2x2 max-pool) -> wave-private fp32 slab [pixel][64 channels] -> 8 bf16 channels (16 B) per lane
    const int cb = n0 + wn * 64;
    float bv[4], dm[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
        bv[bt] = p.bias[cb + 16 * bt + l15];
        if (TRAIN && p.dropmask != nullptr && cb + 16 * bt + l15 < p.Cout) dm[bt] = p.dropmask[(size_t)b * p.Cout + cb + 16 * bt + l15];
    }
    auto fin = [&](float v, int bt) {      // conv + bias -> Dropout2d scale; the ReLU is applied to the packed bf16 pairs in flush()
        v += bv[bt];
        if (TRAIN) v *= dm[bt];
        return v;
    };
    const unsigned relu_floor = p.relu ? 0u : 0x80008000u;
    const int Hy = POOL ? (p.Ho >> 1) : p.Ho;
    const int Wy = POOL ? (p.Wo >> 1) : p.Wo;
    float* slab = reinterpret_cast<float*>((wave & 1) ? stageB : stageA) + (wave >> 1) * (32 * SLAB_P);
    const int prow = lane >> 3, pc8 = (lane & 7) * 8;      // read-back role: pixel row in a group of 8, channel octet
    auto flush = [&](int yy, int xbase) {      // 32 slab rows -> 32 pixels xbase.. of output row yy
        const int nbase = cb + pc8;
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        // ReLU backward of a dgrad launch: the four gate octets of this flush are requested together, in front of its stores. Loaded
        // one by one between the stores, each load was waited for with vmcnt(0) -- its own round trip plus the acknowledgement of the
        // store just issued, 16 times per tile.
        u16x8 gt[4];
        if (TRAIN && p.gate != nullptr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int xx = xbase + g * 8 + prow;
                const bool ok = yy < Hy && xx < Wy && nbase < p.Cout;
                const size_t off = ok ? (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nbase : 0;      // a pixel outside: any valid address, not used
                gt[g] = *reinterpret_cast<const u16x8*>(p.gate + off);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = g * 8 + prow;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + m * SLAB_P + pc8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + m * SLAB_P + pc8 + 4);
            u32x4 ob;
            ob[0] = witw_relu_bf16x2(witw_pack_bf16x2(v0[0], v0[1]), relu_floor);
            ob[1] = witw_relu_bf16x2(witw_pack_bf16x2(v0[2], v0[3]), relu_floor);
            ob[2] = witw_relu_bf16x2(witw_pack_bf16x2(v1[0], v1[1]), relu_floor);
            ob[3] = witw_relu_bf16x2(witw_pack_bf16x2(v1[2], v1[3]), relu_floor);
            bf16x8 o = __builtin_bit_cast(bf16x8, ob);
            const int xx = xbase + m;
            if (yy < Hy && xx < Wy && nbase < p.Cout) {
                const size_t off = (((size_t)b * Hy + yy) * Wy + xx) * p.Cout + nbase;
                if (TRAIN && p.gate != nullptr) {      // zero where the forward's output was <= 0
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if ((gt[g][e] & 0x7fffu) == 0 || (gt[g][e] & 0x8000u)) o[e] = (__bf16)0.f;
                }
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + off));
            }
        }
    };
    if (!POOL) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // tiles 2q, 2q+1: 32 consecutive pixels of one row
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int bt = 0; bt < 4; ++bt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        slab[(16 * i + 4 * kg + r) * SLAB_P + 16 * bt + l15] = fin(acc[2 * q + i][bt][r], bt);
            flush(oy0 + 2 * wm + (q >> 1), ox0 + 32 * (q & 1));
        }
    } else {
        // 2x2 max-pool: vertical partner = tile a + 4 (the row below, same lane and register), horizontal partner = register r ^ 1;
        // the wave's 2 x 64 pixels become one row of 32 pooled pixels: pooled column 8*(a & 3) + 2*kg + e
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bt = 0; bt < 4; ++bt)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float a00 = acc[a][bt][2 * e], a01 = acc[a][bt][2 * e + 1], a10 = acc[a + 4][bt][2 * e], a11 = acc[a + 4][bt][2 * e + 1];
                    const float m4 = fmaxf(fmaxf(a00, a01), fmaxf(a10, a11));
                    slab[(8 * a + 2 * kg + e) * SLAB_P + 16 * bt + l15] = fin(m4, bt);
                    if (TRAIN && p.pool_code != nullptr) {
                        // first position attaining the max, scan order (0,0),(0,1),(1,0),(1,1) as torch's max_pool2d
                        const int yy = (oy0 >> 1) + wm, xx = (ox0 >> 1) + 8 * a + 2 * kg + e, ch = cb + 16 * bt + l15;
                        if (yy < Hy && xx < Wy && ch < p.Cout)
                            p.pool_code[(((size_t)b * Hy + yy) * Wy + xx) * p.Cout + ch] =
                                (unsigned char)((a00 == m4) ? 0 : (a01 == m4) ? 1 : (a10 == m4) ? 2 : 3);
                    }
                }
        flush((oy0 >> 1) + wm, ox0 >> 1);
    }
}

// wpk[nt][kc][tap][g][n][0..7] (bf16) <- w[cout][cin][kh][kw] (fp32, torch KCRS); one thread per 16-B slot.
// transpose_flip != 0 builds the dgrad filter instead: (Cout, Cin) describe the PACKED filter, the source tensor is
// [Cin][Cout][3][3] and w'[co][ci][kh][kw] = w[ci][co][2-kh][2-kw].
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cin,
                                         int n_tiles, int nkc, int TN, int transpose_flip) {
    const size_t total = (size_t)n_tiles * nkc * 9 * 2 * TN;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t t = idx;
    const int n = t % TN; t /= TN;
    const int g = t % 2; t /= 2;
    const int tap = t % 9; t /= 9;
    const int kc = t % nkc; t /= nkc;
    const int nt = (int)t;
    const int kh = tap / 3, kw = tap % 3;
    const int co = nt * TN + n;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = kc * 16 + g * 8 + j;
        float f = 0.f;
        if (co < Cout && ci < Cin)
            f = transpose_flip ? w[(((size_t)ci * Cout + co) * 3 + (2 - kh)) * 3 + (2 - kw)]
                               : w[(((size_t)co * Cin + ci) * 3 + kh) * 3 + kw];
        v[j] = (__bf16)f;
    }
    reinterpret_cast<bf16x8*>(wpk)[idx] = v;
}

// NCHW fp32 [B,C,H,W] -> NHWC bf16 [B,H,W,Cp] (Cp % 16 == 0, extra channels zero). One thread per pixel and 8-channel group: 8
// plane reads that coalesce across the wave (consecutive threads = consecutive pixels of a plane) and ONE 16-byte store -- the
// element-per-thread form wrote 2 bytes per thread and read with a stride of a whole plane (cvig_semantic's training step converts
// its 5 x 128 x 512 inputs with it: 340 us per call at 128 images, round 5: see docs/experiments.md).
__global__ void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int C, int Cp, size_t hw,
                                             size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // over (b, channel group, pixel): pixel fastest
    if (idx >= total) return;
    const int ng = Cp >> 3;
    const size_t r = idx % hw;
    const size_t t = idx / hw;
    const int g = (int)(t % ng);
    const size_t b = t / ng;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = g * 8 + j;
        v[j] = (__bf16)((c < C) ? x[(b * C + c) * hw + r] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(y + (b * hw + r) * Cp + (size_t)g * 8) = v;
}

// Backward of the fused MaxPool2d(2,2) on bf16 NHWC tensors: dy [B,Hp,Wp,C] is routed to the position the forward
// recorded (code = dy*2+dx); dx [B,H,W,C], H >= 2Hp, W >= 2Wp (a dropped odd row / column keeps its memset zeros).
// One thread per pooled pixel and channel octet (16-byte accesses).
__global__ void maxpool2x2_bwd_bf16_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ code,
                                           unsigned short* __restrict__ dx, int Hp, int Wp, int H, int W, int C, size_t total) {
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned char u8x8 __attribute__((ext_vector_type(8)));
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int C8 = C >> 3;
    const int c8 = idx % C8;
    size_t t = idx / C8;
    const int w = t % Wp;
    t /= Wp;
    const int h = t % Hp;
    const size_t b = t / Hp;
    const size_t src = (((b * Hp + h) * Wp + w) * C) + (size_t)c8 * 8;
    const u16x8 g = *reinterpret_cast<const u16x8*>(dy + src);
    const u8x8 k = *reinterpret_cast<const u8x8*>(code + src);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (k[e] == q) ? g[e] : (unsigned short)0;
        *reinterpret_cast<u16x8*>(dx + ((b * H + 2 * h + (q >> 1)) * W + 2 * w + (q & 1)) * C + (size_t)c8 * 8) = o;
    }
}

template <int TN, int SH, bool POOL, int NW>
int launch_bf_nw(ConvBfArgs a, hipStream_t st) {
    a.tiles_y = cdiv(a.Ho, NW);
    const long long sp_total = (long long)a.B * a.tiles_x * a.tiles_y;
    a.n_tiles = cdiv(a.Cout, TN);
    a.sp_per_xcd = (int)((sp_total + 7) / 8);
    const long long grid = a.xcd_map ? 8LL * a.sp_per_xcd * a.n_tiles : sp_total * a.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffLL || sp_total > 0x7fffffffLL) {
        witw_set_error("conv3x3_bf16: grid %lld out of range", grid);
        return WITW_ERR_INVALID;
    }
    a.sp_total = (int)sp_total;
    hipLaunchKernelGGL((conv3x3_nhwc_bf16_kernel<TN, SH, POOL, NW>), dim3((unsigned)grid), dim3(64 * NW), 0, st, a);
    WITW_CHECK_LAUNCH("conv3x3_nhwc_bf16");
    witw_note_variant("conv3x3_nhwc_bf16_kernel<%d,%d,%s,%d>", TN, SH, POOL ? "true" : "false", NW);
    return WITW_OK;
}

// 1: layers that qualify run on the 16x16x32 kernel (default; WITW_BF_S16=0 or witw_conv3x3_bf16_mfma16(0) turn it off)
int g_bf16_mfma16 = -1;
int bf16_mfma16() {
    if (g_bf16_mfma16 < 0) {
        const char* e = getenv("WITW_BF_S16");
        g_bf16_mfma16 = e ? (atoi(e) != 0) : 1;
    }
    return g_bf16_mfma16;
}

template <bool POOL, bool TRAIN>
int launch_bf_s16(ConvBfArgs a, hipStream_t st) {
    a.tiles_y = cdiv(a.Ho, 8);
    const long long sp_total = (long long)a.B * a.tiles_x * a.tiles_y;
    a.n_tiles = cdiv(a.Cout, 128);
    a.sp_per_xcd = (int)((sp_total + 7) / 8);
    const long long grid = a.xcd_map ? 8LL * a.sp_per_xcd * a.n_tiles : sp_total * a.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffLL || sp_total > 0x7fffffffLL) {
        witw_set_error("conv3x3_bf16: grid %lld out of range", grid);
        return WITW_ERR_INVALID;
    }
    a.sp_total = (int)sp_total;
    hipLaunchKernelGGL((conv3x3_bf16_s16_kernel<POOL, TRAIN>), dim3((unsigned)grid), dim3(512), 0, st, a);
    WITW_CHECK_LAUNCH("conv3x3_bf16_s16");
    witw_note_variant("conv3x3_bf16_s16_kernel<%s,%s>", POOL ? "true" : "false", TRAIN ? "true" : "false");
    return WITW_OK;
}

template <int TN, int SH, bool POOL>
int launch_bf(const ConvBfArgs& a, hipStream_t st) {
    const long long big = (long long)cdiv(a.Cout, TN) * a.B * a.tiles_x * cdiv(a.Ho, 8);
    if constexpr (TN == 128 && SH == 1) {
        // the 16x16x32 form: plain inference forward (bf16 NHWC out) of layers large enough for the 8-wave tile
        if (bf16_mfma16() && (a.Ho % 8) == 0 && witw_fills_rounds(big) && !a.out_nchw_f32 && (a.Cout & 7) == 0 && (a.Cin & 31) == 0) {
            if (a.gate || a.dropmask || a.pool_code || a.dil_h) return launch_bf_s16<POOL, true>(a, st);      // training forms
            return launch_bf_s16<POOL, false>(a, st);
        }
    }
    if ((a.Ho % 8) == 0 && witw_fills_rounds(big)) return launch_bf_nw<TN, SH, POOL, 8>(a, st);
    return launch_bf_nw<TN, SH, POOL, 4>(a, st);
}

}  // namespace

#ifdef WITW_BF_STAMPS
unsigned long long* witw_bf16_stamps_ptr = nullptr;
#endif

extern "C" {

// Which MFMA shape the bf16 inference forward uses where both kernels apply (Cout >= 128, stride 1, >= 512 workgroups of 8
// waves, Cin % 32 == 0, bf16 NHWC out, no gate / Dropout2d scale / pool codes): 1 = v_mfma_f32_16x16x32_bf16 (default: the chip
// holds a higher clock on it), 0 = v_mfma_f32_32x32x16_bf16. enable < 0 only queries. Returns the previous setting. The two
// kernels agree to one bf16 unit in the last place (different fp32 summation order).
int witw_conv3x3_bf16_mfma16(int enable) {
    const int prev = bf16_mfma16();
    if (enable >= 0) g_bf16_mfma16 = enable != 0;
    return prev;
}

long long witw_conv3x3_bf16_packed_elems(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return -1;
    const int TN = cout >= 128 ? 128 : 64;
    return (long long)cdiv(cout, TN) * cdiv(cin, 16) * 9 * 2 * TN * 8;
}

int witw_conv3x3_bf16_pack_weights_ex(const float* w_kcrs, void* wpk_bf16, int cout, int cin, int transpose_flip, void* stream) {
    WITW_CHECK_ARG(w_kcrs && wpk_bf16, "bf16 pack_weights: null pointer");
    WITW_CHECK_ARG(cout > 0 && cin > 0, "bf16 pack_weights: bad shape");
    const int TN = cout >= 128 ? 128 : 64;
    const int n_tiles = cdiv(cout, TN), nkc = cdiv(cin, 16);
    const size_t total = (size_t)n_tiles * nkc * 9 * 2 * TN;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_kcrs,
                       (unsigned short*)wpk_bf16, cout, cin, n_tiles, nkc, TN, transpose_flip);
    WITW_CHECK_LAUNCH("bf16 pack_weights");
    return WITW_OK;
}

int witw_conv3x3_bf16_pack_weights(const float* w_kcrs, void* wpk_bf16, int cout, int cin, void* stream) {
    return witw_conv3x3_bf16_pack_weights_ex(w_kcrs, wpk_bf16, cout, cin, 0, stream);
}

int witw_nchw_f32_to_nhwc_bf16(const float* x, void* y_bf16, int B, int C, int H, int W, int Cpad, void* stream) {
    WITW_CHECK_ARG(x && y_bf16, "nchw_f32_to_nhwc_bf16: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && (Cpad % 16) == 0, "nchw_f32_to_nhwc_bf16: bad shape");
    const size_t total = (size_t)B * H * W * (Cpad / 8);
    WITW_CHECK_ARG((total + 255) / 256 <= 0x7fffffffull, "nchw_f32_to_nhwc_bf16: grid too large");
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (__bf16*)y_bf16, C, Cpad, (size_t)H * W, total);
    WITW_CHECK_LAUNCH("nchw_f32_to_nhwc_bf16");
    return WITW_OK;
}

// x NHWC bf16 [B,H,W,Cin] (Cin%16==0) -> y NHWC bf16 [B,Hy,Wy,Cout] (or fp32 NCHW [B,Cout,Hy,Wy] if out_nchw_f32).
// Training form: dropmask [B,Cout] fp32 (Dropout2d scale before the ReLU), gate = bf16 tensor shaped like y (outputs where
// gate <= 0 are zeroed: the ReLU backward of a dgrad launch), dilate_h = x holds (H-1)/2+1 physical rows that stand for H
// zero-interleaved rows (dgrad of a stride-(2,1) layer). H is the LOGICAL input height.
int witw_conv3x3_bf16_fwd_ex(const void* x_bf16, const void* wpk_bf16, const float* bias, const float* dropmask,
                             const void* gate_bf16, void* y, unsigned char* pool_code, int B, int H, int W, int Cin, int Cout,
                             int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, int dilate_h, void* stream) {
    WITW_CHECK_ARG(!pool_code || (pool && (Cout % 8) == 0), "conv3x3_bf16_fwd: pool codes need pool=1 and Cout %% 8 == 0");
    WITW_CHECK_ARG(x_bf16 && wpk_bf16 && bias && y, "conv3x3_bf16_fwd: null pointer");
    WITW_CHECK_ARG(!(gate_bf16 && (pool || out_nchw_f32)), "conv3x3_bf16_fwd: gate with pool / NCHW output unsupported");
    WITW_CHECK_ARG(!(dilate_h && stride_h == 2), "conv3x3_bf16_fwd: dilated input with stride 2 unsupported");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_bf16_fwd: bad shape");
    WITW_CHECK_ARG(Cin > 0 && (Cin % 16) == 0, "conv3x3_bf16_fwd: Cin=%d must be a positive multiple of 16", Cin);
    WITW_CHECK_ARG(stride_h == 1 || stride_h == 2, "conv3x3_bf16_fwd: stride_h=%d unsupported", stride_h);
    WITW_CHECK_ARG(!(pool && stride_h == 2) && !(pool && out_nchw_f32), "conv3x3_bf16_fwd: unsupported pool combination");
    WITW_CHECK_ARG((size_t)H * W * Cin * 2 < 0x80000000ull, "conv3x3_bf16_fwd: image too large for one buffer descriptor");
    ConvBfArgs a;
    a.x = (const unsigned short*)x_bf16; a.wpk = (const unsigned short*)wpk_bf16; a.bias = bias; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Ho = (H + 2 - 3) / stride_h + 1;
    a.Wo = W;
    a.tiles_x = cdiv(a.Wo, TW);
    a.tiles_y = 0;
    a.circ = pad_circular; a.relu = relu; a.out_nchw_f32 = out_nchw_f32;
    a.dropmask = dropmask; a.gate = (const unsigned short*)gate_bf16; a.dil_h = dilate_h ? 1 : 0;
    a.pool_code = pool ? pool_code : nullptr;
#ifdef WITW_BF_STAMPS
    a.stamps = witw_bf16_stamps_ptr;
#endif
    const char* e = getenv("WITW_CONV_XCD");
    a.xcd_map = e ? atoi(e) != 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    // 64 input channels, plain forward: the kernel that keeps the filter in LDS (conv3x3_bf16_wres.hip; bit-identical to the 32x32x16 kernel)
    // (round 5: also the gated form, i.e. the data gradient of a 64-channel layer -- cvig_semantic's layer 2, where layer 0 trains)
    if (stride_h == 1 && !pool && !out_nchw_f32 && !dropmask && !dilate_h && witw_bf16_wres_applies(B, H, W, Cin, Cout))
        return witw_bf16_wres_launch(x_bf16, wpk_bf16, bias, gate_bf16, nullptr, y, B, H, W, Cout, pad_circular, relu, stream);
    if (Cout >= 128) {
        if (stride_h == 2) return launch_bf<128, 2, false>(a, st);
        return pool ? launch_bf<128, 1, true>(a, st) : launch_bf<128, 1, false>(a, st);
    }
    if (stride_h == 2) return launch_bf<64, 2, false>(a, st);
    return pool ? launch_bf<64, 1, true>(a, st) : launch_bf<64, 1, false>(a, st);
}

// The data gradient of a 64-input-channel layer with the ReLU gate as ONE BIT per output (gate_bits [B,H,W,Cout/8] bytes: bit c & 7
// of byte c >> 3 says whether channel c of that pixel passes; written by witw_conv_first2_bf16_fwd_train) instead of a bf16 tensor
// shaped like y: cvig_semantic's layer 2 (model/cvig_semantic.py:301-309: layer 0 trains, so the gradient crosses layer 0's ReLU),
// where the tensor form is the 1.07 GB layer-0 activation. Runs on the weight-resident kernel only: witw_conv3x3_bf16_gatebits_ok
// says whether a shape qualifies (the caller keeps the tensor gate otherwise). Same bits as witw_conv3x3_bf16_fwd_ex with the tensor.
int witw_conv3x3_bf16_gatebits_ok(int B, int H, int W, int Cin, int Cout) { return witw_bf16_wres_applies(B, H, W, Cin, Cout) ? 1 : 0; }

int witw_conv3x3_bf16_fwd_gatebits(const void* x_bf16, const void* wpk_bf16, const float* bias, const void* gate_bits, void* y, int B, int H,
                                   int W, int Cin, int Cout, int pad_circular, int relu, void* stream) {
    WITW_CHECK_ARG(x_bf16 && wpk_bf16 && bias && gate_bits && y, "conv3x3_bf16_fwd_gatebits: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0, "conv3x3_bf16_fwd_gatebits: bad shape");
    WITW_CHECK_ARG(witw_bf16_wres_applies(B, H, W, Cin, Cout),
                   "conv3x3_bf16_fwd_gatebits: B=%d H=%d W=%d Cin=%d Cout=%d does not run on the weight-resident kernel (witw_conv3x3_bf16_gatebits_ok)",
                   B, H, W, Cin, Cout);
    return witw_bf16_wres_launch(x_bf16, wpk_bf16, bias, nullptr, gate_bits, y, B, H, W, Cout, pad_circular, relu, stream);
}

int witw_maxpool2x2_bwd_bf16(const void* dy_bf16, const unsigned char* code, void* dx_bf16, int B, int Hp, int Wp, int H, int W,
                             int C, void* stream) {
    WITW_CHECK_ARG(dy_bf16 && code && dx_bf16, "maxpool2x2_bwd_bf16: null pointer");
    WITW_CHECK_ARG(B > 0 && Hp > 0 && Wp > 0 && C > 0 && (C % 8) == 0 && H >= 2 * Hp && W >= 2 * Wp, "maxpool2x2_bwd_bf16: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if ((H > 2 * Hp || W > 2 * Wp) && hipMemsetAsync(dx_bf16, 0, 2 * (size_t)B * H * W * C, st) != hipSuccess) {
        witw_set_error("maxpool2x2_bwd_bf16: memset failed");
        return WITW_ERR_LAUNCH;
    }
    const size_t total = (size_t)B * Hp * Wp * (C / 8);
    hipLaunchKernelGGL(maxpool2x2_bwd_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const unsigned short*)dy_bf16, code, (unsigned short*)dx_bf16, Hp, Wp, H, W, C, total);
    WITW_CHECK_LAUNCH("maxpool2x2_bwd_bf16");
    return WITW_OK;
}

int witw_conv3x3_bf16_fwd(const void* x_bf16, const void* wpk_bf16, const float* bias, void* y, int B, int H, int W, int Cin,
                          int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, void* stream) {
    return witw_conv3x3_bf16_fwd_ex(x_bf16, wpk_bf16, bias, nullptr, nullptr, y, nullptr, B, H, W, Cin, Cout, stride_h, pad_circular, relu,
                                    pool, out_nchw_f32, 0, stream);
}

}  // extern "C"
