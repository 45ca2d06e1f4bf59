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
//   B  layer 0 on the 10 x 34 positions of the layer-2 input tile: 11 M-tiles of 32 positions x 2 N-tiles of 32 channels (22
//      units over the 8 waves), accumulators started from the bias, MFMA with the
//      FILTER as the A operand so that a lane ends up with 4 consecutive channels of one position = one 8-byte LDS write into
//      the layer-2 operand image [channel group of 8][position]; positions outside the picture are layer 2's zero padding
//   C  layer 2: 4 K chunks x 9 taps of v_mfma_f32_32x32x16_bf16, M-tile = 2 rows x 16 columns (a 2x2 pooling window lies inside
//      one lane's registers), one M-tile x 2 N-tiles per wave, operands by hand-issued ds_read_b128 three steps ahead with
//      counted lgkmcnt waits, same accumulation order as
//      conv3x3_nhwc_bf16_kernel (chunk-major, tap-minor): bit-identical to the unfused launches
//   D  bias + ReLU + 2x2 max, through a wave-private 1 KB slab to 16-byte NHWC stores.
#include "common.h"
#include "lds_frag.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int F2T = 512;                 // 8 waves, two per SIMD (the layer-0 phase is VALU-heavy: two waves keep a SIMD's vector issue full)
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
    unsigned char* pool_code; // TRAIN: arg-max codes of the fused max-pool, [B,H/2,W/2,64] (dy*2+dx, first maximum in scan order)
    unsigned char* gate_bits; // TRAIN: layer 0's ReLU gate, one bit per output: [B,H,W,8] bytes, bit c & 7 of byte c >> 3 = channel c > 0
    int B, C, H, W;
    int tiles_x, tiles_y;
    int n_tiles;
    int circ;
};

__device__ unsigned long long f2_stamps[2][8];       // WITW_F2_STAMPS=1 diagnostic: phase ticks of waves 0 and 7, third tile of workgroup 0

// TRAIN: what the backward of a training step needs instead of the two activations (cvig_semantic: layer 0 trains, so the gradient
// goes back through both layers, model/cvig_semantic.py:301-309): the arg-max codes of the max-pool and layer 0's ReLU gate as one
// bit per output (64 bits per pixel = 67 MB at 128 images against the 1.07 GB bf16 map), both produced in phase D -- the codes from
// the accumulators, the bits from the layer-2 operand image, which stays in LDS until the next tile's phase B.
template <int CW, bool REC, bool TRAIN>
__global__ __launch_bounds__(F2T, 1) void conv_first2_bf16_kernel(First2Args p) {
    static_assert(CW == 4 || CW == 8, "4 or 8 bf16 per raw pixel");
    constexpr int NMF = (CW == 4) ? 3 : 5;
    typedef typename std::conditional<CW == 4, u32x2, u32x4>::type pix_t;
    __shared__ u32x4 a_s[8 * APOS];                 // 61,440 B: layer-0 output tile, [group of 8 channels][row][pitch 48]
    __shared__ u32x4 w_s[4 * 9 * 2 * 64];           // 73,728 B: layer-2 filter, resident for every tile of this workgroup
    __shared__ pix_t raw_s[RH * RW];                // 3.4 / 6.9 KB
    __shared__ u32x4 slab_s[8 * 64];                // 8 KB: one 1 KB output slab per wave
    __shared__ unsigned char cslab_s[TRAIN ? 8 * 512 : 16];      // TRAIN: one 512-byte code slab per wave
    __shared__ float b0_s[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hq = lane >> 5;
    const size_t plane = (size_t)p.H * p.W;
    const int Hy = p.H >> 1, Wy = p.W >> 1;

    // ---- once per workgroup: layer-2 filter -> LDS, layer-0 filter fragments and layer-2 bias -> registers
    for (int s = tid; s < 4 * 9 * 2 * 64; s += F2T) w_s[s] = p.wpk2[s];
    if (tid < 64) b0_s[tid] = p.bias0[tid];
    // layer 0 is cut into 22 units (M-tile of 32 positions, N-tile of 32 channels); wave w takes units w, w+8, w+16, whose
    // N-tile is w & 1 for all three: one N-tile's filter fragments and bias per wave
    const int nt0 = wave & 1;
    u32x4 aw[NMF];                                  // layer-0 filter as the MFMA A operand: row = channel nt0*32 + l31, k half = hq
#pragma unroll
    for (int i = 0; i < NMF; ++i) aw[i] = p.wf0[(i * 2 + hq) * 64 + nt0 * 32 + l31];
    float b2[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b2[nt] = p.bias2[nt * 32 + l31];
    f32x16 b0r;                                     // layer-0 bias of this lane's 16 channels (register 4j+e: channel nt0*32 + 8j + 4hq + e)
#pragma unroll
    for (int r = 0; r < 16; ++r) b0r[r] = p.bias0[nt0 * 32 + 8 * (r >> 2) + 4 * hq + (r & 3)];

    // raw pixels of this thread (tile-relative; 432 pixels, one per thread), fetched one tile ahead. Buffer loads over the image's C
    // planes: a channel >= C, a row / column outside the picture and a tile past the end are out-of-range offsets or an empty
    // descriptor (zeros) -- straight-line code, so the wait in front of the conversion below counts exactly these loads.
    constexpr int NRAW = (RH * RW + F2T - 1) / F2T;
    constexpr unsigned OOR = 0xfffffff0u;
    float rv[NRAW][CW];
    const unsigned img_bytes = (unsigned)p.C * (unsigned)plane * 4u;       // < 2^31 (checked by the launcher)
    const int tiles_img = p.tiles_x * p.tiles_y;
    auto fetch_raw = [&](int tile) {
        const bool any = tile < p.n_tiles;
        const int tt = any ? tile : 0;
        const int b = tt / tiles_img, rem = tt - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const unsigned char* img = reinterpret_cast<const unsigned char*>(p.x) + (size_t)b * img_bytes;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, any ? img_bytes : 0u, 0x00020000);
#pragma unroll
        for (int k = 0; k < NRAW; ++k) {
            const int s = tid + k * F2T;
            const int rr = s / RW, rc = s - rr * RW;
            const int gr = ty * TH2 - 2 + rr;
            int gc = tx * TW2 - 2 + rc;
            bool ok = s < RH * RW && gr >= 0 && gr < p.H;
            if (p.circ) {
                if (p.W >= RW) gc = gc < 0 ? gc + p.W : (gc >= p.W ? gc - p.W : gc);      // gc in [-2, W + 33]: one wrap
                else { gc %= p.W; if (gc < 0) gc += p.W; }
            } else {
                ok = ok && gc >= 0 && gc < p.W;
            }
            const unsigned off = ok ? (unsigned)(gr * p.W + gc) * 4u : OOR;
#pragma unroll
            for (int ch = 0; ch < CW; ++ch)       // plane ch rides in soffset, which the range check does NOT cover (it sees voffset only):
                // a missing plane (ch >= C, wave-uniform) is made out of range through the voffset, so that it reads zeros, not the next image
                rv[k][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ch < p.C ? off : OOR,
                                                                                           (unsigned)ch * (unsigned)plane * 4u, 0));
        }
    };
    auto raw_to_lds = [&]() {                       // phase A: this thread's raw pixel -> LDS as bf16
#pragma unroll
        for (int k = 0; k < NRAW; ++k) {
            const int s = tid + k * F2T;
            if (s < RH * RW) {
                __bf16 v[CW];
#pragma unroll
                for (int ch = 0; ch < CW; ++ch) v[ch] = (__bf16)rv[k][ch];
                raw_s[s] = __builtin_bit_cast(pix_t, v);
            }
        }
    };

    // layer-2 roles: wave = (row pair, column half); M-tile = 2 rows x 16 columns, lane l31 -> (row l31 >> 4, column l31 & 15)
    const int prow = wave >> 1, chalf = wave & 1;
    const int a_lane = ((2 * prow + (l31 >> 4)) * APITCH + 16 * chalf + (l31 & 15));      // + (kh * APITCH + kw) per tap

    // Phase A of the first tile runs here, of every later tile behind its predecessor's layer-2 loop: one place inside the loop
    // where the prefetched registers are consumed, always behind the same sequence of memory operations, so that its wait counts
    // exactly the prefetch loads -- reached from two paths of different depth (loop top) it is vmcnt(0), which also waits for the
    // previous tile's output store: a store round trip per tile.
    fetch_raw(blockIdx.x);
    raw_to_lds();
    // everything loaded so far (filter fragments, biases) has landed: said with the builtin, so that the compiler's counter
    // bookkeeping enters the loop clean -- otherwise the first use of such a register INSIDE the loop carries a vmcnt(0) in every
    // iteration, i.e. a wait for the prefetch loads issued just before it
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0), expcnt / lgkmcnt untouched
    int iter = 0;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x, ++iter) {
        const bool rec = REC && blockIdx.x == 0 && iter == 2 && lane == 0 && (wave == 0 || wave == 7);
        auto stamp = [&](int k) { if (rec) f2_stamps[wave == 7][k] = __builtin_amdgcn_s_memtime(); };
        stamp(0);
        const int b = tile / tiles_img, rem = tile - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * TH2, ox0 = tx * TW2;
        const bool border = oy0 == 0 || oy0 + TH2 + 1 > p.H || (!p.circ && (ox0 == 0 || ox0 + TW2 + 1 > p.W));      // wave-uniform

        stamp(1);
        __syncthreads();          // raw tile visible; every wave has left the previous tile's layer-2 loop (a_s is free)
        stamp(2);
        fetch_raw(tile + (int)gridDim.x);

        // ---- B: layer 0 on the 10 x 34 positions: 11 M-tiles x 2 N-tiles = 22 units over the 8 waves, three per wave (the last
        // one of waves 6 and 7 repeats unit 2 and is not written). All raw operands of the three units are read first, then the
        // three accumulation chains advance in turn (one chain alone would wait out every MFMA's latency and every LDS read).
        // The accumulator starts from the bias, as conv3x3_first_bf16_kernel's does: one max and half a convert per value.
        {
            constexpr int NU = 3;
            int qr[NU], qc[NU];
            bool live[NU];
            u32x4 pv[NU][NMF];
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                const int mt = min((wave >> 1) + 4 * k, NMT0 - 1);
                const int q = mt * 32 + l31;                // position index in the 10 x 34 tile
                live[k] = q < AH * AW && (wave >> 1) + 4 * k < NMT0;
                qr[k] = q < AH * AW ? q / AW : 0;
                qc[k] = q < AH * AW ? q - qr[k] * AW : 0;
                const int rbase = qr[k] * RW + qc[k];       // raw pixel of tap (0,0)
#pragma unroll
                for (int i = 0; i < NMF; ++i) {
                    if constexpr (CW == 4) {
                        const int tA = 4 * i + 2 * hq, tB = tA + 1;
                        u32x2 pa = raw_s[rbase + ((tA < 9) ? (tA / 3) * RW + tA % 3 : 0)];
                        u32x2 pb = raw_s[rbase + ((tB < 9) ? (tB / 3) * RW + tB % 3 : 0)];
                        if (tA >= 9) pa = (u32x2){0u, 0u};
                        if (tB >= 9) pb = (u32x2){0u, 0u};
                        pv[k][i] = (u32x4){pa[0], pa[1], pb[0], pb[1]};
                    } else {
                        const int t = 2 * i + hq;
                        pv[k][i] = raw_s[rbase + ((t < 9) ? (t / 3) * RW + t % 3 : 0)];
                        if (t >= 9) pv[k][i] = (u32x4){0u, 0u, 0u, 0u};
                    }
                }
            }
            f32x16 acc0[NU];
#pragma unroll
            for (int k = 0; k < NU; ++k) acc0[k] = b0r;
#pragma unroll
            for (int i = 0; i < NMF; ++i)
#pragma unroll
                for (int k = 0; k < NU; ++k)       // D[channel][position]: filter rows x pixel columns
                    acc0[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[i]), __builtin_bit_cast(bf16x8, pv[k][i]),
                                                                      acc0[k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                // position inside the picture? (outside: layer 2's zero padding, or the wrapped column under circular padding);
                // only tiles on the picture's edge need the test
                bool outside = false;
                if (border) {
                    const int gy = oy0 - 1 + qr[k], gx = ox0 - 1 + qc[k];
                    outside = gy < 0 || gy >= p.H;
                    if (!p.circ) outside = outside || gx < 0 || gx >= p.W;
                }
                if (live[k]) {
                    unsigned char* dst = reinterpret_cast<unsigned char*>(a_s) + ((size_t)(qr[k] * APITCH + qc[k])) * 16 + 8 * hq;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {           // registers 4j..4j+3 = channels nt0*32 + 8j + 4hq + {0..3}
                        u32x2 ob;                               // ReLU on the packed pairs (common.h)
                        ob[0] = witw_relu_bf16x2(witw_pack_bf16x2(acc0[k][4 * j], acc0[k][4 * j + 1]), 0u);
                        ob[1] = witw_relu_bf16x2(witw_pack_bf16x2(acc0[k][4 * j + 2], acc0[k][4 * j + 3]), 0u);
                        if (outside) ob = (u32x2){0u, 0u};
                        *reinterpret_cast<u32x2*>(dst + (size_t)(nt0 * 4 + j) * (APOS * 16)) = ob;
                    }
                }
            }
        }
        stamp(3);
        __syncthreads();          // layer-2 input tile complete
        stamp(4);

        // ---- C: layer 2, 36 (chunk, tap) steps of 2 MFMAs per wave (1 M-tile x 2 N-tiles). Operand reads are inline asm (left to
        // the scheduler each ds_read sinks to just in front of its MFMA and every MFMA pair waits on lgkmcnt(0)): asm volatile
        // statements keep their order, reads are issued in the order their MFMAs need them (fa, fb0 -> MFMA 1; fb1 -> MFMA 2)
        // PF steps ahead, and LDS returns in order, so an MFMA waits with lgkmcnt(reads issued after the last one it needs).
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        constexpr int PF = 3, NB = PF + 1;
        u32x4 fa[NB], fb[NB][2];
        const unsigned abase = lds_address(a_s) + (unsigned)(hq * APOS + a_lane) * 16u;
        const unsigned wbase = lds_address(w_s) + (unsigned)(hq * 64 + l31) * 16u;
        auto a_addr = [&](int step) {
            const int kc = step / 9, tap = step - kc * 9;
            const int kh = tap / 3, kw = tap - kh * 3;
            return abase + (unsigned)(2 * kc * APOS + kh * APITCH + kw) * 16u;
        };
        auto w_addr = [&](int step, int nt) { return wbase + (unsigned)(step * 128 + nt * 32) * 16u; };
        int issued = 0;
        auto issue = [&](int step, int which) {
            const int bq = step % NB;
            if (which == 0) fa[bq] = lds_read128(a_addr(step));
            else fb[bq][which - 1] = lds_read128(w_addr(step, which - 1));
            ++issued;
        };
#pragma unroll
        for (int st = 0; st < PF; ++st)
#pragma unroll
            for (int which = 0; which < 3; ++which) issue(st, which);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int bq = step % NB;
            const bool more = step + PF < 36;
            lds_wait(issued - (3 * step + 2), fa[bq], fb[bq][0]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[bq]), __builtin_bit_cast(bf16x8, fb[bq][0]), acc[0], 0, 0, 0);
            if (more) {
                issue(step + PF, 0);
                issue(step + PF, 1);
            }
            lds_wait(issued - (3 * step + 3), fb[bq][1]);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[bq]), __builtin_bit_cast(bf16x8, fb[bq][1]), acc[1], 0, 0, 0);
            if (more) issue(step + PF, 2);
        }

        stamp(5);
        // ---- A (next tile): raw_s was last read in phase B, a barrier ago. Before D, so that the tile's output store is issued
        // AFTER these registers' loads have been waited for (the other way round the last wait is vmcnt(0): the store's round trip).
        // The wait is stated here for every wave: inside raw_to_lds it sits behind a branch that the waves without a raw pixel
        // skip, and a load left "pending" on that path costs a vmcnt(0) at the loop top, i.e. behind the store.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        raw_to_lds();
        // ---- D: bias + ReLU + 2x2 max-pool -> slab [pooled column 0..7][64 channels] bf16 -> 16-byte stores
        // register r <-> pixel m = (r&3) + 8*(r>>2) + 4*hq of the M-tile, m = 16*row + column: the window of pooled column
        // jp = (r&3)/2 + 4*((r>>2)&1) + 2*hq is registers {r, r+1, r+8, r+9} (r&3 in {0,2}, r < 8)
        __bf16* slab = reinterpret_cast<__bf16*>(slab_s + wave * 64);
        unsigned char* cslab = cslab_s + (TRAIN ? wave * 512 : 0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = (jj & 1) * 2 + (jj >> 1) * 4;
                const float m4 = fmaxf(fmaxf(acc[nt][r], acc[nt][r + 1]), fmaxf(acc[nt][r + 8], acc[nt][r + 9]));
                const int jp = (jj & 1) + 4 * (jj >> 1) + 2 * hq;
                slab[jp * 64 + nt * 32 + l31] = (__bf16)fmaxf(m4 + b2[nt], 0.f);
                if (TRAIN)      // first position attaining the max, scan order (0,0),(0,1),(1,0),(1,1) as torch's max_pool2d (conv3x3_bf16.hip)
                    cslab[jp * 64 + nt * 32 + l31] = (unsigned char)((acc[nt][r] == m4) ? 0 : (acc[nt][r + 1] == m4) ? 1 : (acc[nt][r + 8] == m4) ? 2 : 3);
            }
        if (TRAIN) __builtin_amdgcn_wave_barrier();      // (compiler only: the other lanes' slab writes stay in front of the reads)
        {
            const int jp = lane >> 3, c8 = (lane & 7) * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + jp * 64 + c8);
            const int py = (oy0 >> 1) + prow, px = (ox0 >> 1) + 8 * chalf + jp;
            if (py < Hy && px < Wy) {
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.y + (((size_t)b * Hy + py) * Wy + px) * 64 + c8));
                if (TRAIN) {
                    const u32x2 cv = *reinterpret_cast<const u32x2*>(cslab + jp * 64 + c8);
                    *reinterpret_cast<u32x2*>(p.pool_code + (((size_t)b * Hy + py) * Wy + px) * 64 + c8) = cv;
                }
            }
        }
        if (TRAIN) {
            // layer 0's ReLU gate of the tile's 8 x 32 interior positions out of the layer-2 operand image (a_s: post-ReLU bf16, 8
            // channels per 16-byte slot, [group][row][pitch 48]): wave w takes row w, lane l31 column l31, the lower half-wave
            // channel groups 0-3 and the upper one groups 4-7 (a ds_read_b128's 16-lane groups stay inside one half: 16 consecutive
            // slots, conflict-free); the halves are joined by one cross-lane move and lanes 0-31 store 8 bytes each -- 256
            // contiguous bytes per row. a_s is not written again before the next tile's first barrier.
            // the image holds post-ReLU values (witw_relu_bf16x2: every half is in [0x0000, 0x7fff]), so "> 0" is "!= 0": a packed
            // unsigned min with 1 turns a dword into its two gate bits (positions 0 and 16), shift-ors gather a group's four
            // dwords (positions 0,2,4,6 / 16,18,20,22) and one shift folds the upper ones in between: ~10 vector instructions per
            // 8 channels (the compare-and-select form took 32)
            // (v_pk_min_u16 by hand: written as __builtin_elementwise_min on a 2 x u16 view of sv[e], hipcc 7.2 folded the four
            // dwords of a slot into its first one -- a one-dword load and t = m | m << 2 | m << 4 | m << 6)
            const unsigned one2 = 0x00010001u;
            unsigned wbits = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 sv = a_s[(4 * hq + g) * APOS + (wave + 1) * APITCH + (l31 + 1)];
                unsigned m0, m1, m2, m3;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(m0) : "v"(sv[0]), "v"(one2));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(m1) : "v"(sv[1]), "v"(one2));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(m2) : "v"(sv[2]), "v"(one2));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(m3) : "v"(sv[3]), "v"(one2));
                const unsigned t = m0 | (m1 << 2) | (m2 << 4) | (m3 << 6);
                wbits |= ((t & 0x55u) | ((t >> 15) & 0xAAu)) << (8 * g);
            }
            const unsigned other = (unsigned)__shfl_xor((int)wbits, 32, 64);
            const int oy = oy0 + wave, ox = ox0 + l31;
            if (hq == 0 && oy < p.H && ox < p.W)
                *reinterpret_cast<u32x2*>(p.gate_bits + (((size_t)b * p.H + oy) * p.W + ox) * 8) = (u32x2){wbits, other};
        }
        stamp(6);
    }
}

}  // namespace

extern "C" {

static int first2_launch(const float* x, const void* wf0, const float* bias0, const void* wpk2, const float* bias2, void* y,
                         unsigned char* pool_code, unsigned char* gate_bits, int B, int C, int H, int W, int pad_circular, void* stream) {
    const bool train = pool_code != nullptr;
    First2Args a;
    a.x = x; a.wf0 = (const u32x4*)wf0; a.bias0 = bias0; a.wpk2 = (const u32x4*)wpk2; a.bias2 = bias2; a.y = (unsigned short*)y;
    a.pool_code = pool_code; a.gate_bits = gate_bits;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.tiles_x = cdiv(W, TW2); a.tiles_y = cdiv(H, TH2);
    const long long n_tiles = (long long)B * a.tiles_x * a.tiles_y;
    WITW_CHECK_ARG(n_tiles < 0x7fffffffLL && (unsigned long long)C * H * W * 4 < 0x80000000ull, "conv_first2_bf16: tensor too large");
    a.n_tiles = (int)n_tiles;
    a.circ = pad_circular;
    const int n_cu = witw_cu_count();        // persistent workgroups, one per CU (150 KB of LDS each)
    const unsigned grid = (unsigned)(a.n_tiles < n_cu ? a.n_tiles : n_cu);
    const bool rec = !train && getenv("WITW_F2_STAMPS") != nullptr && a.n_tiles >= 3 * (int)grid;      // diagnostic, synchronous
    if (rec && C > 4)
        hipLaunchKernelGGL((conv_first2_bf16_kernel<8, true, false>), dim3(grid), dim3(F2T), 0, (hipStream_t)stream, a);
    else if (train && C <= 4)
        hipLaunchKernelGGL((conv_first2_bf16_kernel<4, false, true>), dim3(grid), dim3(F2T), 0, (hipStream_t)stream, a);
    else if (train)
        hipLaunchKernelGGL((conv_first2_bf16_kernel<8, false, true>), dim3(grid), dim3(F2T), 0, (hipStream_t)stream, a);
    else if (C <= 4)
        hipLaunchKernelGGL((conv_first2_bf16_kernel<4, false, false>), dim3(grid), dim3(F2T), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv_first2_bf16_kernel<8, false, false>), dim3(grid), dim3(F2T), 0, (hipStream_t)stream, a);
    if (rec && C > 4) {
        (void)hipDeviceSynchronize();
        unsigned long long h[2][8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(f2_stamps), sizeof(h)) == hipSuccess)
            for (int w = 0; w < 2; ++w)
                fprintf(stderr, "conv_first2 wave %d, third tile (ticks): raw->LDS %llu, barrier %llu, layer 0 %llu, barrier %llu, layer 2 %llu, "
                                "pool+store %llu, total %llu\n", w ? 7 : 0, h[w][1] - h[w][0], h[w][2] - h[w][1], h[w][3] - h[w][2],
                        h[w][4] - h[w][3], h[w][5] - h[w][4], h[w][6] - h[w][5], h[w][6] - h[w][0]);
    }
    WITW_CHECK_LAUNCH("conv_first2_bf16");
    witw_note_variant("conv_first2_bf16_kernel<%d,%s>", C <= 4 ? 4 : 8, train ? "train" : "infer");
    return WITW_OK;
}

// x NCHW fp32 [B,C<=8,H,W] -> y NHWC bf16 [B,H/2,W/2,64] = MaxPool2(ReLU(conv2(ReLU(conv0(x))))), bf16 operands / fp32 accumulate.
// wf0 / bias0: witw_conv3x3_first_pack(round_bf16 = 1) image and bias of the first conv; wpk2 / bias2: witw_conv3x3_bf16_pack_weights
// image (64 -> 64) and bias of the second. Bit-identical to witw_conv3x3_first_fwd(out_bf16 = 1) followed by
// witw_conv3x3_bf16_fwd(relu, pool).
int witw_conv_first2_bf16_fwd(const float* x, const void* wf0, const float* bias0, const void* wpk2, const float* bias2, void* y,
                              int B, int C, int H, int W, int pad_circular, void* stream) {
    WITW_CHECK_ARG(x && wf0 && bias0 && wpk2 && bias2 && y, "conv_first2_bf16: null pointer");
    WITW_CHECK_ARG(B > 0 && C >= 1 && C <= 8 && H >= 2 && W >= 2, "conv_first2_bf16: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    return first2_launch(x, wf0, bias0, wpk2, bias2, y, nullptr, nullptr, B, C, H, W, pad_circular, stream);
}

// The training form: the same y, plus what the backward through both layers needs in place of the two activations -- pool_code
// [B,H/2,W/2,64] (arg-max of each pooling window, dy*2+dx, as witw_conv3x3_bf16_fwd_ex writes them: input of
// witw_maxpool2x2_bwd_bf16) and gate_bits [B,H,W,8] bytes (bit c & 7 of byte c >> 3: layer 0's output channel c at that pixel is
// > 0: the gate of witw_conv3x3_bf16_fwd_gatebits). Layer 2's own ReLU gate is y > 0. H and W must be even.
int witw_conv_first2_bf16_fwd_train(const float* x, const void* wf0, const float* bias0, const void* wpk2, const float* bias2, void* y,
                                    unsigned char* pool_code, unsigned char* gate_bits, int B, int C, int H, int W, int pad_circular,
                                    void* stream) {
    WITW_CHECK_ARG(x && wf0 && bias0 && wpk2 && bias2 && y && pool_code && gate_bits, "conv_first2_bf16_train: null pointer");
    WITW_CHECK_ARG(B > 0 && C >= 1 && C <= 8 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0,
                   "conv_first2_bf16_train: bad shape B=%d C=%d H=%d W=%d (H, W even)", B, C, H, W);
    return first2_launch(x, wf0, bias0, wpk2, bias2, y, pool_code, gate_bits, B, C, H, W, pad_circular, stream);
}

}  // extern "C"
