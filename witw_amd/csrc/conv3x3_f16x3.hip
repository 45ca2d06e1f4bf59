// 3x3 convolution with fp32-grade accuracy on the fp16 MFMA (gfx950): every fp32 value x is carried as TWO fp16 numbers,
// hi = fp16(x) and lo = fp16(x - hi) (22 significand bits; the matrix core honours fp16 subnormals — probed on MI355X),
// and a product x*w is formed as hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16 with fp32 accumulation (the lo*lo
// term, 2^-22 relative, is dropped). Three MFMAs of the 2.5 PF/s pipe per fp32-equivalent product against one of the
// 157 TF/s fp32 pipe. A CPU emulation of the whole encoder stays within 2e-5 of fp64 — the distance the fp32 MFMA path
// has too — i.e. inside the 1e-4 bound the reference goldens are held to (tests/test_f16x3_gpu.py).
//
// FOV_DSM encoder of the reference (model/cvig_fov.py:256-294), inference. Structure = conv3x3_bf16.hip. What differs:
//   * activations are NHWC "split-fp16": per pixel and 8 channels 16 B of hi followed by 16 B of lo (4 B per value like
//     fp32). A K chunk = 8 channels = 32 B per pixel; LDS slot group 0 = hi, group 1 = lo;
//   * per tap one MFMA takes A = [x_hi | x_lo] (the two lane halves read the two groups, as the bf16 kernel reads its
//     two channel groups) against B = [w_hi | w_hi] (both halves read the same slab slot): x_hi*w_hi + x_lo*w_hi.
//     The cross term x_hi*w_lo pairs TWO TAPS in one MFMA: A = [x_hi(tap a) | x_hi(tap b)], B = [w_lo(a) | w_lo(b)].
//     9 + 5 = 14 MFMA steps per 8-channel chunk (the 9th tap pairs with zeros);
//   * packed filter per chunk: 9 slots w_hi(tap) + 5 x 2 slots w_lo(tap pair) = 19 slots of TN x 16 B.
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 64;
constexpr int IW = TW + 2;
constexpr int NSLOT = 19;       // weight slab slots per chunk: 9 hi taps + 5 lo tap pairs x 2
constexpr int NSTEP = 14;       // MFMA steps per chunk: 9 taps + 5 tap pairs
#ifndef WITW_HX_SPREAD
#define WITW_HX_SPREAD 5        // steps over which the staging pieces of a chunk are issued
#endif

__device__ __forceinline__ i32x4 raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// one wave instruction of LDS-DMA (see conv3x3_bf16.hip): lane l moves 16 B from rs[voff_l + soff] to LDS lds_addr + 16*l
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

struct ConvHxArgs {
    const unsigned short* x;   // [B,H,W,Cin/8,2,8] split-fp16 NHWC (Cin % 8 == 0)
    const unsigned short* wpk; // packed fp16: [n_tile][cin/8][19 slots][TN][8]
    const float* bias;         // [n_tiles*TN] fp32 (zero padded)
    void* y;                   // split-fp16 NHWC [B,Hy,Wy,Cout/8,2,8] or fp32 NCHW [B,Cout,Hy,Wy]
    int B, H, W, Cin, Cout;
    int Ho, Wo;
    int tiles_x, tiles_y;
    int circ, relu, out_nchw_f32;
    int n_tiles, sp_total, sp_per_xcd, xcd_map;
    const float* dropmask;       // nullptr, or [B,Cout] Dropout2d scale applied to conv + bias before the ReLU
    const unsigned short* gate;  // nullptr, or a split-fp16 tensor shaped like y: outputs where its value <= 0 are zeroed (ReLU backward)
    int dil_h;                   // 1: input rows are zero-interleaved (logical row 2i = physical row i): dgrad of a stride-(2,1) conv
    unsigned char* pool_code;    // nullptr, or [B,Hy,Wy,Cout] arg-max position (dy*2+dx) of the fused 2x2 max pool (Cout % 8 == 0)
    int* overflow;               // nullptr, or a device flag set to 1 when an output leaves the fp16 range (|v| > 65504: its hi part is inf)
};

template <int TN, int SH, bool POOL, int NW>
__global__ __launch_bounds__(64 * NW) void conv3x3_nhwc_f16x3_kernel(ConvHxArgs p) {
    constexpr int TH = NW;
    constexpr int NTHREADS = 64 * NW;
    constexpr int IH = (TH - 1) * SH + 3;
    constexpr int IN_S = 2 * IH * IW;           // 16-B slots of one input stage (hi and lo planes of 8 channels)
    constexpr int IN_P = (IN_S + 63) / 64 * 64;
    constexpr int W_S = NSLOT * TN;             // 16-B slots of one weight stage
    constexpr int STAGE_S = IN_P + W_S;
    constexpr int NIN = (IN_S + NTHREADS - 1) / NTHREADS;
    constexpr int NWT_D = (W_S / 64 + NW - 1) / NW;
    constexpr int WGM = (TN == 128) ? NW / 2 : NW;
    constexpr int WM = (2 * TH) / WGM;
    constexpr int WN = 2;
    constexpr unsigned OOR = 0x80000000u;
    static_assert(W_S % 64 == 0, "weight stage must be whole wave instructions");
    static_assert(STAGE_S * 16 >= (NW / 2) * 32 * 64 * 4, "a stage must hold the epilogue slabs of half the waves");
    static_assert(2 * STAGE_S * 16 + 16 <= 160 * 1024, "two stages must fit the LDS");

    __shared__ u32x4 stageA[STAGE_S];
    __shared__ u32x4 stageB[STAGE_S + 1];
    u32x4* const dummy_slot = stageB + STAGE_S;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave) & (NW - 1);
    const int l31 = lane & 31, hq = lane >> 5;

    // block -> (n tile, spatial tile): XCD-aware order, see conv3x3.hip
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
    const int nkc = p.Cin >> 3;

    // ---- staging: per-image / per-weight-tile buffer resources, fixed per-thread byte offsets, K-chunk advance in the
    // scalar offset (32 B per pixel per chunk: 8 channels x (hi, lo))
    const int Hp = p.dil_h ? (p.H - 1) / 2 + 1 : p.H;      // physical rows of the input
    const size_t img_halves = (size_t)Hp * p.W * p.Cin * 2;
    __amdgpu_buffer_rsrc_t in_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * img_halves), 0, (unsigned)(img_halves * 2), 0x00020000);
    const i32x4 w_rd = raw_rsrc(reinterpret_cast<const u32x4*>(p.wpk) + (size_t)ntile * nkc * W_S, (unsigned)nkc * W_S * 16u);
    unsigned gin[NIN];          // slot s -> pixel s/2, plane s%2 (a pixel's 32 B of a chunk load as one segment)
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int s = tid + i * NTHREADS;
        const int pix = s >> 1, q = s & 1;
        const int r = pix / IW, c = pix - r * IW;
        int gr = oy0 * SH - 1 + r;
        int gc = ox0 - 1 + c;
        bool ok = s < IN_S && gr >= 0 && gr < p.H;
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
        gin[i] = ok ? (unsigned)((((size_t)gr * p.W + gc) * (2 * p.Cin) + q * 8) * 2) : OOR;
    }
    u32x4 rin[NIN];
    const unsigned lane16 = (unsigned)lane * 16u;

    constexpr int PIECES = NIN + NWT_D;
    auto stage_piece = [&](int kc, u32x4* in_s, int pc) {
        const unsigned in_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(in_s));
        if (pc < NIN) {
            rin[pc] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, gin[pc], (unsigned)kc * 32u, 0);
        } else {
            const int j = wave_u + NW * (pc - NIN);
            if (NWT_D * NW == W_S / 64 || j < W_S / 64)
                dma16(w_rd, in_lds + (unsigned)(IN_P + j * 64) * 16u, lane16, (unsigned)kc * W_S * 16u + (unsigned)j * 1024u);
        }
    };
    auto stage_commit = [&](u32x4* in_s) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int s = tid + i * NTHREADS;
            u32x4* dst = (NIN * NTHREADS == IN_S || s < IN_S) ? in_s + (s & 1) * (IH * IW) + (s >> 1) : dummy_slot;
            *dst = rin[i];
        }
    };
    auto stage_wait = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };     // vmcnt(0)

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
    // A-operand bases: tap steps read plane hq (hi | lo) at the tap's pixel; pair steps read plane 0 (hi) at the pixel
    // of tap 2i (lanes 0-31) or tap 2i+1 (lanes 32-63)
    int abase[WM], abase_hi[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        abase_hi[mt] = trow[mt] * SH * IW + tcol[mt] + l31;
        abase[mt] = abase_hi[mt] + hq * (IH * IW);
    }
    int pairoff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int ta = 2 * i, tb = (2 * i + 1 < 9) ? 2 * i + 1 : 8;      // tap 9 does not exist: its w_lo slot is zero
        pairoff[i] = hq ? (tb / 3) * IW + tb % 3 : (ta / 3) * IW + ta % 3;
    }
    const int wbase = wn * 64 + l31;        // B operand of a tap step: both lane halves read w_hi
    const int wbase_p = hq * TN + wbase;    // B operand of a pair step: half hq reads w_lo of its tap

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    u32x4 fa[2][WM], fb[2][WN];
    auto read_frags = [&](int set, const u32x4* in_s, const u32x4* w_s, int step) {
        if (step < 9) {
            const int kh = step / 3, kw = step - kh * 3;
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) fa[set][mt] = in_s[abase[mt] + kh * IW + kw];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) fb[set][nt] = w_s[step * TN + wbase + nt * 32];
        } else {
            const int i = step - 9;
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) fa[set][mt] = in_s[abase_hi[mt] + pairoff[i]];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) fb[set][nt] = w_s[(9 + 2 * i) * TN + wbase_p + nt * 32];
        }
    };
    auto mfma_step = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][mt]),
                                                                    __builtin_bit_cast(f16x8, fb[set][nt]), acc[mt][nt], 0, 0, 0);
    };

#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) stage_piece(0, stageA, pc);
    stage_commit(stageA);
    stage_wait();
    __syncthreads();
    read_frags(0, stageA, stageA + IN_P, 0);

    // one K chunk: 14 MFMA steps out of stage `in_s` while chunk kc+1 moves into stage `in_n`; the fragments of step
    // s+1 are read under the MFMAs of step s, the chunk's barrier sits in front of the last step
    auto chunk = [&](const u32x4* in_s, u32x4* in_n, int kc) {
        const int kn = (kc + 1 < nkc) ? kc + 1 : kc;
        const u32x4* w_s = in_s + IN_P;
#pragma unroll
        for (int st = 0; st < NSTEP - 1; ++st) {
            read_frags((st + 1) & 1, in_s, w_s, st + 1);
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc)
                if (pc * WITW_HX_SPREAD / PIECES == st) stage_piece(kn, in_n, pc);
            if (st == 8) stage_commit(in_n);
            mfma_step(st & 1);
            if (st != 0 && st != 8) {       // one fragment read per MFMA
#pragma unroll
                for (int i = 0; i < WM + WN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < WM * WN - (WM + WN); ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
        stage_wait();
        __syncthreads();
        read_frags(0, in_n, in_n + IN_P, 0);        // NSTEP is even: the next chunk starts on fragment set 0 again
        mfma_step((NSTEP - 1) & 1);
    };
    static_assert((NSTEP & 1) == 0, "fragment sets alternate per step");
    for (int kc = 0; kc < nkc; kc += 2) {
        chunk(stageA, stageB, kc);
        if (kc + 1 < nkc) chunk(stageB, stageA, kc + 1);
    }
    __syncthreads();      // the slabs below reuse the stages

    // ---- epilogue: bias, ReLU, optional 2x2 max pool; the fp32 tile is transposed through a wave-private LDS slab and
    // leaves split into (hi, lo) fp16 as 2 x 16 B per lane (8 channels of a pixel), or as the fp32 NCHW embedding
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
    // a split-fp16 value is > 0 iff its hi part is (lo only refines it; hi = 0 means the value itself was 0 or tiny positive:
    // a saved ReLU output below the smallest fp16 subnormal counts as closed, like an exact zero)
    auto gate_open = [](unsigned short g) { return (g & 0x7fffu) != 0 && !(g & 0x8000u); };
    const int Hy = POOL ? (p.Ho >> 1) : p.Ho;
    const int Wy = POOL ? (p.Wo >> 1) : p.Wo;
    _Float16* const yh = reinterpret_cast<_Float16*>(p.y);
    // element (pixel index pix, channel c) of a split-fp16 tensor with C channels: hi at this offset, lo 8 halves further
    auto split_off = [&](size_t pix, int c) { return (pix * p.Cout + (size_t)(c & ~7)) * 2 + (c & 7); };
    bool too_big_e = false;
    auto emit = [&](float v, int nt, int yy, int xx) {
        v = fin(v, nt);
        if (yy < Hy && xx < Wy && nch[nt] < p.Cout) {
            if (p.out_nchw_f32) {
                reinterpret_cast<float*>(p.y)[(((size_t)b * p.Cout + nch[nt]) * Hy + yy) * Wy + xx] = v;
            } else {
                const size_t o = split_off(((size_t)b * Hy + yy) * Wy + xx, nch[nt]);
                if (p.gate != nullptr && !gate_open(p.gate[o])) v = 0.f;
                too_big_e = too_big_e || !(fabsf(v) <= 65504.f);
                const _Float16 hi = (_Float16)v;
                yh[o] = hi;
                yh[o + 8] = (_Float16)(v - (float)hi);
            }
        }
    };
    // 8 channels (two float4) of one pixel -> 16 B hi + 16 B lo, adjacent
    bool too_big = false;       // some value of this lane left the fp16 range
    auto store_split8 = [&](f32x4 v0, f32x4 v1, size_t pix, int nbase) {
#pragma unroll
        for (int e = 0; e < 4; ++e) too_big = too_big || !(fabsf(v0[e]) <= 65504.f) || !(fabsf(v1[e]) <= 65504.f);
        if (p.gate != nullptr) {
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            const u16x8 gt = *reinterpret_cast<const u16x8*>(p.gate + (pix * p.Cout + nbase) * 2);     // hi plane of the gate
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!gate_open(gt[e])) v0[e] = 0.f;
                if (!gate_open(gt[4 + e])) v1[e] = 0.f;
            }
        }
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (_Float16)v0[e];
            hi[4 + e] = (_Float16)v1[e];
            lo[e] = (_Float16)(v0[e] - (float)hi[e]);
            lo[4 + e] = (_Float16)(v1[e] - (float)hi[4 + e]);
        }
        f16x8* dst = reinterpret_cast<f16x8*>(yh + (pix * p.Cout + nbase) * 2);
        __builtin_nontemporal_store(hi, dst);
        __builtin_nontemporal_store(lo, dst + 1);
    };

    float* slab = reinterpret_cast<float*>((wave & 1) ? stageB : stageA) + (wave >> 1) * (32 * 64);
    const int prow = lane >> 3, pc8 = (lane & 7) * 8;   // read-back role: pixel row in a group of 8, channel octet
    if (!POOL && !p.out_nchw_f32 && (p.Cout & 7) == 0) {
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
                const int xx = ox0 + tcol[mt] + m;
                if (yy < Hy && xx < Wy && nbase < p.Cout) store_split8(v0, v1, ((size_t)b * Hy + yy) * Wy + xx, nbase);
            }
        }
    } else if (!POOL) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    emit(acc[mt][nt][r], nt, oy0 + trow[mt], ox0 + tcol[mt] + (r & 3) + 8 * (r >> 2) + 4 * hq);
    } else if ((p.Cout & 7) == 0) {
        // fused 2x2 max-pool: 16 pooled pixels x 64 channels per M-tile pair
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
                if (yy < Hy && xb + pc < Wy && nbase < p.Cout) store_split8(v0, v1, ((size_t)b * Hy + yy) * Wy + xb + pc, nbase);
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
                        emit(fmaxf(v0, v1), nt, (oy0 + trow[mtA]) >> 1, ((ox0 + tcol[mtA]) >> 1) + 4 * g + 2 * hq + e);
                    }
        }
    }
    // a value beyond the fp16 range (or a NaN) cannot be carried as hi + lo: raise the caller's flag instead of passing infinities on
    if (p.overflow != nullptr && __any((too_big || too_big_e) ? 1 : 0) && lane == 0) atomicOr(p.overflow, 1);
}

// wpk[nt][kc][slot][n][0..7] (fp16) <- w[cout][cin][kh][kw] (fp32, torch KCRS); one thread per 16-B slot.
// slot s < 9: hi part of tap s; slot 9 + 2i + h: lo part of tap 2i + h (zeros for tap 9), channels 8kc..8kc+7.
// transpose_flip != 0: the dgrad filter: (Cout, Cin) describe the PACKED filter, the source is [Cin][Cout][3][3] and
// w'[co][ci][kh][kw] = w[ci][co][2-kh][2-kw].
__global__ void pack_weights_f16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cin,
                                          int n_tiles, int nkc, int TN, int transpose_flip) {
    const size_t total = (size_t)n_tiles * nkc * NSLOT * TN;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t t = idx;
    const int n = t % TN; t /= TN;
    const int slot = t % NSLOT; t /= NSLOT;
    const int kc = t % nkc; t /= nkc;
    const int nt = (int)t;
    const bool lo = slot >= 9;
    const int tap = lo ? slot - 9 : slot;           // lo slots: 9 + tap (tap 9 = zeros)
    const int co = nt * TN + n;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = kc * 8 + j;
        float f = 0.f;
        if (co < Cout && ci < Cin && tap < 9)
            f = transpose_flip ? w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : w[((size_t)co * Cin + ci) * 9 + tap];
        const _Float16 hi = (_Float16)f;
        v[j] = lo ? (_Float16)(f - (float)hi) : hi;
    }
    reinterpret_cast<f16x8*>(wpk)[idx] = v;
}

// NCHW fp32 [B,C,H,W] -> split-fp16 NHWC [B,H,W,Cp/8,2,8] (Cp % 8 == 0, extra channels zero); one thread per (pixel, octet)
__global__ void nchw_f32_to_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int C, int Cp, size_t hw,
                                         size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int C8 = Cp >> 3;
    const int c8 = idx % C8;
    const size_t t = idx / C8;
    const size_t b = t / hw, r = t - b * hw;
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        const float f = (c < C) ? x[(b * C + c) * hw + r] : 0.f;
        hi[j] = (_Float16)f;
        lo[j] = (_Float16)(f - (float)hi[j]);
    }
    f16x8* dst = reinterpret_cast<f16x8*>(y + (t * Cp + (size_t)c8 * 8) * 2);
    dst[0] = hi;
    dst[1] = lo;
}

// split-fp16 NHWC -> fp32 NHWC (tests, and hand-over to fp32 consumers)
__global__ void split_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int C, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (pixel, channel)
    if (idx >= total) return;
    const int c = idx % C;
    const size_t pix = idx / C;
    const _Float16* xh = reinterpret_cast<const _Float16*>(x);
    const size_t o = (pix * C + (size_t)(c & ~7)) * 2 + (c & 7);
    y[idx] = (float)xh[o] + (float)xh[o + 8];
}

// Backward of the fused MaxPool2d(2,2) on split-fp16 tensors: dy [B,Hp,Wp,C/8,2,8] is routed (both planes) to the position
// the forward recorded (code uint8 [B,Hp,Wp,C]); dx [B,H,W,C/8,2,8], H >= 2Hp, W >= 2Wp (a dropped odd row / column keeps its
// zeros). One thread per pooled pixel and channel octet.
__global__ void maxpool2x2_bwd_split_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ code,
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
    const size_t pp = (b * Hp + h) * Wp + w;
    const u16x8 ghi = *reinterpret_cast<const u16x8*>(dy + ((pp * C8 + c8) * 2) * 8);
    const u16x8 glo = *reinterpret_cast<const u16x8*>(dy + ((pp * C8 + c8) * 2 + 1) * 8);
    const u8x8 k = *reinterpret_cast<const u8x8*>(code + pp * C + (size_t)c8 * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u16x8 ohi, olo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ohi[e] = (k[e] == q) ? ghi[e] : (unsigned short)0;
            olo[e] = (k[e] == q) ? glo[e] : (unsigned short)0;
        }
        u16x8* dst = reinterpret_cast<u16x8*>(dx + ((((b * H + 2 * h + (q >> 1)) * W + 2 * w + (q & 1)) * C8 + c8) * 2) * 8);
        dst[0] = ohi;
        dst[1] = olo;
    }
}

template <int TN, int SH, bool POOL, int NW>
int launch_hx_nw(ConvHxArgs a, hipStream_t st) {
    a.tiles_y = cdiv(a.Ho, NW);
    const long long sp_total = (long long)a.B * a.tiles_x * a.tiles_y;
    a.n_tiles = cdiv(a.Cout, TN);
    a.sp_per_xcd = (int)((sp_total + 7) / 8);
    const long long grid = a.xcd_map ? 8LL * a.sp_per_xcd * a.n_tiles : sp_total * a.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffLL || sp_total > 0x7fffffffLL) {
        witw_set_error("conv3x3_f16x3: grid %lld out of range", grid);
        return WITW_ERR_INVALID;
    }
    a.sp_total = (int)sp_total;
    hipLaunchKernelGGL((conv3x3_nhwc_f16x3_kernel<TN, SH, POOL, NW>), dim3((unsigned)grid), dim3(64 * NW), 0, st, a);
    WITW_CHECK_LAUNCH("conv3x3_nhwc_f16x3");
    witw_note_variant("conv3x3_nhwc_f16x3_kernel<%d,%d,%s,%d>", TN, SH, POOL ? "true" : "false", NW);
    return WITW_OK;
}

template <int TN, int SH, bool POOL>
int launch_hx(const ConvHxArgs& a, hipStream_t st) {
    const long long big = (long long)cdiv(a.Cout, TN) * a.B * a.tiles_x * cdiv(a.Ho, 8);
    if ((a.Ho % 8) == 0 && witw_fills_rounds(big)) return launch_hx_nw<TN, SH, POOL, 8>(a, st);
    return launch_hx_nw<TN, SH, POOL, 4>(a, st);
}

}  // namespace

extern "C" {

long long witw_conv3x3_f16x3_packed_elems(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return -1;
    const int TN = cout >= 128 ? 128 : 64;
    return (long long)cdiv(cout, TN) * cdiv(cin, 8) * NSLOT * TN * 8;
}

int witw_conv3x3_f16x3_pack_weights_ex(const float* w_kcrs, void* wpk_f16, int cout, int cin, int transpose_flip, void* stream) {
    WITW_CHECK_ARG(w_kcrs && wpk_f16, "f16x3 pack_weights: null pointer");
    WITW_CHECK_ARG(cout > 0 && cin > 0, "f16x3 pack_weights: bad shape");
    const int TN = cout >= 128 ? 128 : 64;
    const int n_tiles = cdiv(cout, TN), nkc = cdiv(cin, 8);
    const size_t total = (size_t)n_tiles * nkc * NSLOT * TN;
    hipLaunchKernelGGL(pack_weights_f16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_kcrs,
                       (unsigned short*)wpk_f16, cout, cin, n_tiles, nkc, TN, transpose_flip);
    WITW_CHECK_LAUNCH("f16x3 pack_weights");
    return WITW_OK;
}

int witw_conv3x3_f16x3_pack_weights(const float* w_kcrs, void* wpk_f16, int cout, int cin, void* stream) {
    return witw_conv3x3_f16x3_pack_weights_ex(w_kcrs, wpk_f16, cout, cin, 0, stream);
}

int witw_nchw_f32_to_split_f16(const float* x, void* y_split, int B, int C, int H, int W, int Cpad, void* stream) {
    WITW_CHECK_ARG(x && y_split, "nchw_f32_to_split_f16: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && (Cpad % 8) == 0, "nchw_f32_to_split_f16: bad shape");
    const size_t total = (size_t)B * H * W * (Cpad / 8);
    hipLaunchKernelGGL(nchw_f32_to_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)y_split, C, Cpad, (size_t)H * W, total);
    WITW_CHECK_LAUNCH("nchw_f32_to_split_f16");
    return WITW_OK;
}

int witw_split_f16_to_f32(const void* x_split, float* y, long long pixels, int C, void* stream) {
    WITW_CHECK_ARG(x_split && y, "split_f16_to_f32: null pointer");
    WITW_CHECK_ARG(pixels > 0 && C > 0 && (C % 8) == 0, "split_f16_to_f32: bad shape");
    const size_t total = (size_t)pixels * C;
    hipLaunchKernelGGL(split_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x_split, y, C, total);
    WITW_CHECK_LAUNCH("split_f16_to_f32");
    return WITW_OK;
}

// x split-fp16 NHWC [B,H,W,Cin/8,2,8] (Cin % 8 == 0) -> y split-fp16 NHWC [B,Hy,Wy,Cout/8,2,8] (Cout % 8 == 0) or the fp32 NCHW
// embedding [B,Cout,Hy,Wy] (out_nchw_f32).
// Training form (the trainable layers' forward and the dgrad launches of a backward): dropmask [B,Cout] fp32 (Dropout2d scale
// before the ReLU), gate = split-fp16 tensor shaped like y (outputs where it is <= 0 are zeroed), dilate_h = x holds
// (H-1)/2+1 physical rows standing for H zero-interleaved rows (dgrad of a stride-(2,1) layer; H is the logical height).
int witw_conv3x3_f16x3_fwd_ex(const void* x_split, const void* wpk_f16, const float* bias, const float* dropmask,
                              const void* gate_split, void* y, unsigned char* pool_code, int* overflow_flag, int B, int H, int W,
                              int Cin, int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, int dilate_h,
                              void* stream) {
    WITW_CHECK_ARG(!pool_code || (pool && (Cout % 8) == 0), "conv3x3_f16x3_fwd: pool codes need pool=1 and Cout %% 8 == 0");
    WITW_CHECK_ARG(x_split && wpk_f16 && bias && y, "conv3x3_f16x3_fwd: null pointer");
    WITW_CHECK_ARG(!(gate_split && (pool || out_nchw_f32)), "conv3x3_f16x3_fwd: gate with pool / NCHW output unsupported");
    WITW_CHECK_ARG(!(dilate_h && stride_h == 2), "conv3x3_f16x3_fwd: dilated input with stride 2 unsupported");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_f16x3_fwd: bad shape");
    WITW_CHECK_ARG(Cin > 0 && (Cin % 8) == 0, "conv3x3_f16x3_fwd: Cin=%d must be a positive multiple of 8", Cin);
    WITW_CHECK_ARG(out_nchw_f32 || (Cout % 8) == 0, "conv3x3_f16x3_fwd: a split-fp16 output needs Cout %% 8 == 0 (Cout=%d)", Cout);
    WITW_CHECK_ARG(stride_h == 1 || stride_h == 2, "conv3x3_f16x3_fwd: stride_h=%d unsupported", stride_h);
    WITW_CHECK_ARG(!(pool && stride_h == 2) && !(pool && out_nchw_f32), "conv3x3_f16x3_fwd: unsupported pool combination");
    WITW_CHECK_ARG((size_t)H * W * Cin * 4 < 0x80000000ull, "conv3x3_f16x3_fwd: image too large for one buffer descriptor");
    ConvHxArgs a;
    a.x = (const unsigned short*)x_split; a.wpk = (const unsigned short*)wpk_f16; a.bias = bias; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Ho = (H + 2 - 3) / stride_h + 1;
    a.Wo = W;
    a.tiles_x = cdiv(a.Wo, TW);
    a.tiles_y = 0;
    a.circ = pad_circular; a.relu = relu; a.out_nchw_f32 = out_nchw_f32;
    a.dropmask = dropmask; a.gate = (const unsigned short*)gate_split; a.dil_h = dilate_h ? 1 : 0;
    a.pool_code = pool ? pool_code : nullptr;
    a.overflow = overflow_flag;
    const char* e = getenv("WITW_CONV_XCD");
    a.xcd_map = e ? atoi(e) != 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (Cout >= 128) {
        if (stride_h == 2) return launch_hx<128, 2, false>(a, st);
        return pool ? launch_hx<128, 1, true>(a, st) : launch_hx<128, 1, false>(a, st);
    }
    if (stride_h == 2) return launch_hx<64, 2, false>(a, st);
    return pool ? launch_hx<64, 1, true>(a, st) : launch_hx<64, 1, false>(a, st);
}

int witw_maxpool2x2_bwd_split(const void* dy_split, const unsigned char* code, void* dx_split, int B, int Hp, int Wp, int H, int W,
                              int C, void* stream) {
    WITW_CHECK_ARG(dy_split && code && dx_split, "maxpool2x2_bwd_split: null pointer");
    WITW_CHECK_ARG(B > 0 && Hp > 0 && Wp > 0 && C > 0 && (C % 8) == 0 && H >= 2 * Hp && W >= 2 * Wp, "maxpool2x2_bwd_split: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if ((H > 2 * Hp || W > 2 * Wp) && hipMemsetAsync(dx_split, 0, 4 * (size_t)B * H * W * C, st) != hipSuccess) {
        witw_set_error("maxpool2x2_bwd_split: memset failed");
        return WITW_ERR_LAUNCH;
    }
    const size_t total = (size_t)B * Hp * Wp * (C / 8);
    hipLaunchKernelGGL(maxpool2x2_bwd_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const unsigned short*)dy_split, code, (unsigned short*)dx_split, Hp, Wp, H, W, C, total);
    WITW_CHECK_LAUNCH("maxpool2x2_bwd_split");
    return WITW_OK;
}

int witw_conv3x3_f16x3_fwd(const void* x_split, const void* wpk_f16, const float* bias, void* y, int B, int H, int W, int Cin,
                           int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, void* stream) {
    return witw_conv3x3_f16x3_fwd_ex(x_split, wpk_f16, bias, nullptr, nullptr, y, nullptr, nullptr, B, H, W, Cin, Cout, stride_h, pad_circular, relu,
                                     pool, out_nchw_f32, 0, stream);
}

}  // extern "C"
