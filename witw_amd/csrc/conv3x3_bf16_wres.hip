// 3x3 convolution, bf16 inference, 64 input channels, filter RESIDENT in LDS (gfx950).
//
// Layer 5 of the FOV_DSM trunk (features[5] of VGG16: Conv2d(64 -> 128, 3x3, pad 1) + ReLU on the 64 x 256 map,
// model/cvig_fov.py:261-262) has K = 9 x 64 only: in the tiled kernels (conv3x3_bf16.hip) a workgroup's K loop is 4 chunks long
// and the per-tile prologue (first weight / input stage) + epilogue (accumulators -> LDS -> stores) take 40 % of its time
// (0.38 of the matrix peak against 0.69 on the 512-channel layers). Here a persistent workgroup owns one block of 64 output
// channels, keeps that block's whole filter (4 chunks x 9 taps x 64 channels x 16 bf16 = 73.7 KB) in LDS for all of its tiles
// and only stages the input tile of each output tile -- the layer-2 phase of conv_first2_bf16.hip with the input read from HBM
// instead of being recomputed. The workgroups of the other channel blocks walk the same tiles in the same order on the same XCD,
// so the second read of an input tile is an L2 hit.
//
// Round 4: TWO TEAMS of four waves (one wave of each team per SIMD) on two 8 x 16 tiles half a tile apart in time, as in
// conv_first2_bf16.hip. Round 3's form ran all eight waves through MFMA loop -> epilogue -> input-to-LDS together (stamps per 8 x 32
// tile: MFMA loop 3.3 k cycles for the older wave of a SIMD and 5.5 k for the younger, epilogue 1.4-1.6 k, barriers and LDS fill
// 1 k: 8.0 k against 4.6 k of matrix time). Now, between two workgroup barriers, one team is in its M phase (72 MFMAs per wave
// back to back + the next tile's global loads, one between MFMA steps) while the other is in its V phase (prefetched input ->
// LDS, then bias / ReLU / slab / stores of its previous tile).
//
// Per team and tile (8 x 16 pixels x 64 channels): V: the next tile's input, prefetched into registers during the previous M
// phase, -> LDS [channel group of 8][row][pitch 24]; bias + ReLU -> bf16 -> a wave-private slab [8 pixels][64 channels] (8-byte
// writes) -> 16-byte NHWC stores, one row of the wave's four at a time. M: 4 K chunks x 9 taps of v_mfma_f32_32x32x16_bf16 per
// wave (M-tile = 4 rows x 8 columns: 32 pixels x 64 channels), the FILTER as the A operand so that a lane ends up with 4
// consecutive channels of one pixel, operands by hand-issued ds_read_b128 three steps ahead (lds_frag.h); accumulation order
// chunk-major, tap-minor as in conv3x3_nhwc_bf16_kernel: BIT-IDENTICAL to that kernel.
#include "common.h"
#include "lds_frag.h"
#include <stdlib.h>

#ifndef WITW_WRES_PF
#define WITW_WRES_PF 3             // (chunk, tap) steps the layer-2 operand reads run ahead of their MFMAs
#endif
#ifndef WITW_WRES_DMA
#define WITW_WRES_DMA 1         // 1: the input tile by LDS-DMA into a pixel-major, XOR-swizzled image; 0: through registers + ds_write_b128
#endif
#ifndef WITW_WRES_DIRECT
#define WITW_WRES_DIRECT 0      // 1: the epilogue stores from registers (v_permlane32_swap forms 16-byte channel octets); 0: through the LDS slab.
                                // Measured (same box): in-kernel cycles per iteration 6.7 k -> 5.6 k, wall time 0.266-0.279 -> 0.270-0.275 ms with
                                // plain stores (0.449 with non-temporal ones: the four 32-byte pieces of a line leave L2 one by one): no gain, off
#endif
#ifndef WITW_WRES_DIAG
#define WITW_WRES_DIAG 0        // diagnostic builds (wrong results): 1 = no waits on the operand reads, 2 = no operand reads, 4 = no V-phase work, 8 = no input -> LDS writes, 16 = no slab traffic in the epilogue
#endif
#ifndef WITW_WRES_PRIO
#define WITW_WRES_PRIO 1        // s_setprio 1 for the M phase: the matrix-bound wave of a SIMD wins the issue arbitration
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WRT = 512;                          // 8 waves: two teams of four
constexpr int WTEAM = 256;
constexpr int WTH = 8, WTW = 16;                  // output tile of a team
constexpr int WAH = WTH + 2, WAW = WTW + 2;       // input tile: 10 x 18 positions
constexpr int WPITCH = 24;                        // LDS row pitch in 16-byte slots (= 8 mod 16: the four rows of an M-tile on disjoint banks)
constexpr int WPOS = WAH * WPITCH + 1;            // 241 slots per channel group: the 8 groups of one pixel (8 neighbouring lanes of the staging
                                                  // write) start 4 banks apart -- ds_write_b128 serves 8 contiguous lanes per cycle, bank (a/4) % 32
constexpr int WNIN = (WAH * WAW * 8 + WTEAM - 1) / WTEAM;      // 16-byte input chunks per thread and tile (6)
// WITW_WRES_DMA: the team's input image is PIXEL-major -- pixel p = row * 18 + column holds its eight 16-byte channel groups in
// slots 8p .. 8p+7, group g in slot 8p + (g ^ f), f = ((column >> 1) & 1) | ((row & 3) << 1) -- so that one LDS-DMA instruction copies
// 1 KB of contiguous global memory (8 pixels x 128 bytes, whole cache lines; each lane fetches the 16 bytes that belong at its LDS
// slot) and the MFMA operand reads (32 pixels of a 4 x 8 M-tile, one group) stay free of bank conflicts: within every 16-lane
// group of ds_read_b128 the pairs (column parity, f) are distinct for all nine taps (checked exhaustively). No register transit,
// no ds_write: the switch-off builds of the register form showed its 6 ds_write_b128 per wave and tile costing 12 % of the kernel
// (they back the LDS queue up under the other team's operand reads).
constexpr int DPIECES = (WAH * WAW * 128 + 1023) / 1024;      // 23 DMA pieces of 1 KB per tile
constexpr int DIMG = DPIECES * 1024;                           // bytes of a team's image
constexpr int DPW = (DPIECES + 3) / 4;                         // pieces per wave (6)
constexpr int SLAB_PITCH = 144;                   // bytes per pixel of the output slab (128 + 16: 8-byte writes of 16 lanes on distinct banks)
constexpr int SLAB_BYTES = 16 * SLAB_PITCH;       // two rows of the M-tile: 16 pixels

struct WresArgs {
    const u32x4* x;           // NHWC bf16 [B,H,W,64] as 16-byte channel groups
    const u32x4* wpk;         // conv3x3_bf16 packing [n_tile][4 chunks][9 taps][2 groups][TN][8 bf16]
    const float* bias;        // [>= Cout]
    unsigned short* y;        // NHWC bf16 [B,H,W,Cout]
    const unsigned short* gate;      // GATE instantiation: bf16 tensor shaped like y; outputs where gate <= 0 are zeroed (the ReLU backward of a dgrad launch)
    const unsigned char* gate_bits;  // GATE = 2: the same gate as ONE BIT per output, [B,H,W,Cout/8] bytes, bit c & 7 of byte c >> 3 = channel c passes
                                     // (written by the training form of conv_first2_bf16_kernel: 67 MB instead of a 1.07 GB activation)
    int B, H, W, Cout;
    int tiles_x, tiles_y, n_sp;      // spatial (team) tiles per row / column of an image, in all
    int n_cb;                 // blocks of 64 output channels
    int w_tn;                 // TN of the packing (64 or 128)
    int q_per_xcd;            // tile-pair walkers per XCD and channel block
    int circ, relu;
};

__device__ __forceinline__ void wres_wave_sync() {      // one wave's LDS traffic is processed in issue order: drain the counter, pin the compiler
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// global -> LDS, 16 B per lane, 1 KB of contiguous LDS per wave instruction at lds_addr (M0); out-of-range lanes write zeros
__device__ __forceinline__ void wres_dma16(i32x4 rs, unsigned lds_addr, unsigned voff) {
    // m0 is written here and is NOT on the clobber list: it is a reserved register for LLVM's AMDGPU back end (clang warns "clobber
    // list contains reserved registers: m0 ... undefined behaviour" when it is listed), which keeps no value live in it across
    // instructions and re-sets it right before each of its own uses (LDS-DMA builtins, movrel, sendmsg). ADVICE r04.
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rs)
                 : "memory");
}

__device__ unsigned long long wres_stamps[2][8];     // WITW_WRES_STAMPS=1 diagnostic: phase ticks of waves 0 and 4 (one per team), third iteration of workgroup 0

template <bool REC, int GATE>      // GATE: 0 none, 1 = bf16 tensor shaped like y, 2 = one bit per output (gate_bits)
__global__ __launch_bounds__(WRT, 1) void conv3x3_bf16_wres_kernel(WresArgs p) {
    __shared__ __attribute__((aligned(1024))) u32x4 a_s[WITW_WRES_DMA ? 2 * DIMG / 16 : 2 * 8 * WPOS];      // per team the input tile (47,104 B; register form 61,696)
    __shared__ u32x4 w_s[4 * 9 * 2 * 64];           // 73,728 B: this channel block's filter
    __shared__ u32x4 slab_s[8 * SLAB_BYTES / 16];   // 18,432 B: one [16 pixels][64 channels] bf16 slab per wave
    __shared__ f32x4 bias_s[16];                    // this channel block's bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, wt = wave & 3, tt = tid & (WTEAM - 1);
    const int l31 = lane & 31, hq = lane >> 5;
    u32x4* const a_t = a_s + team * (WITW_WRES_DMA ? DIMG / 16 : 8 * WPOS);

    // workgroup -> (XCD, channel block, walker): block i runs on XCD i % 8; the n_cb workgroups of a walker share its tiles
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = slot % p.n_cb, walker = slot / p.n_cb;
    const int pair_first = xcd + 8 * walker, pair_step = 8 * p.q_per_xcd;
    const int n_pairs = (p.n_sp + 1) >> 1;
    const int n_it = pair_first < n_pairs ? (n_pairs - 1 - pair_first) / pair_step + 1 : 0;
    auto tile_of = [&](int it) { return 2 * (pair_first + it * pair_step) + team; };

    // ---- once: filter block -> LDS, bias -> registers
    {
        const int per_tile = p.w_tn / 64;           // channel blocks per packed n-tile
        const u32x4* wsrc = p.wpk + (size_t)(cb / per_tile) * (4 * 9 * 2) * p.w_tn + (cb % per_tile) * 64;
        for (int s = tid; s < 4 * 9 * 2 * 64; s += WRT) w_s[s] = wsrc[(size_t)(s >> 6) * p.w_tn + (s & 63)];
    }
    if (tid < 64) reinterpret_cast<float*>(bias_s)[tid] = p.bias[cb * 64 + tid];      // read back by the epilogue (32 values per lane: kept out of the M phase's registers)

    // this thread's input chunks (tile-relative, the same for every tile): 8 lanes = the 128 bytes of one pixel, a wave = 8 neighbouring
    // pixels of a row (whole cache lines). Per tile only a scalar base is added; the chunks on the tile's rim carry flags, and what
    // a flag means for this tile (padding = out-of-range offset -> the load returns zeros; circular wrap = +-one row length) is
    // decided on scalars: a dozen vector instructions per tile and thread, no branch.
    constexpr unsigned F_TOP = 1, F_BOT = 2, F_LEFT = 4, F_RIGHT = 8, F_DEAD = 16;
    constexpr unsigned OOR = 0xfffffff0u;           // a buffer offset outside any image
    unsigned in_meta[WNIN];                         // slot in a_t (low 16 bits) | rim flags << 16
    unsigned in_rel[WNIN];                          // byte offset from the tile's first halo pixel (row -1, column -1)
#pragma unroll
    for (int k = 0; k < WNIN; ++k) {
        const int s = tt + k * WTEAM;
        const int g = s & 7, pos = s >> 3;
        const bool live = pos < WAH * WAW;
        const int r = live ? pos / WAW : 0, c = live ? pos - r * WAW : 0;
        in_rel[k] = ((unsigned)r * (unsigned)p.W + (unsigned)c) * 128u + (unsigned)g * 16u;
        in_meta[k] = (unsigned)(g * WPOS + r * WPITCH + c) |
                     (((live ? 0u : F_DEAD) | (r == 0 ? F_TOP : 0u) | (r == WAH - 1 ? F_BOT : 0u) | (c == 0 ? F_LEFT : 0u) | (c == WAW - 1 ? F_RIGHT : 0u)) << 16);
    }
    const int tiles_img = p.tiles_x * p.tiles_y;
    const unsigned img_bytes = (unsigned)p.H * (unsigned)p.W * 128u;      // < 2^31 (checked by the caller)
    const unsigned row_bytes = (unsigned)p.W * 128u;
    u32x4 rv[WNIN];
    // WITW_WRES_DMA: wave wt issues pieces wt * DPW .. of the team's image; lane -> slot s = 64 * piece + lane = pixel s >> 3, physical
    // group slot s & 7, which holds channel group (s & 7) ^ f(pixel). dmeta = byte offset from the tile's first halo pixel | rim
    // flags in the low 4 bits (the offset is a multiple of 16); bit 31 = no pixel (the image's tail) or no piece.
    unsigned dmeta[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int piece = wt * DPW + j;
        const int sl = piece * 64 + lane;
        const int px = sl >> 3, k = sl & 7;
        const bool live = piece < DPIECES && px < WAH * WAW;
        const int r = live ? px / WAW : 0, c = live ? px - r * WAW : 0;
        const int g = k ^ (((c >> 1) & 1) | ((r & 3) << 1));
        dmeta[j] = live ? ((((unsigned)r * (unsigned)p.W + (unsigned)c) * 128u + (unsigned)g * 16u) |
                           (r == 0 ? F_TOP : 0u) | (r == WAH - 1 ? F_BOT : 0u) | (c == 0 ? F_LEFT : 0u) | (c == WAW - 1 ? F_RIGHT : 0u))
                        : 0x80000000u;
    }
    i32x4 d_rs = {0, 0, 0, 0x00020000};
    auto dma_setup = [&](int t) {                   // the scalar part of fetch_setup with the descriptor as four SGPRs for the asm
        const bool any = t < p.n_sp;
        const int tl = any ? t : 0;
        const int b = tl / tiles_img;
        const unsigned long long a = (unsigned long long)(reinterpret_cast<const unsigned char*>(p.x) + (size_t)b * img_bytes);
        d_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        d_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
        d_rs[2] = __builtin_amdgcn_readfirstlane(any ? (int)img_bytes : 0);
    };
    // Buffer loads, a tile past the end = empty descriptor: straight-line code, so the wait in front of the LDS write counts exactly
    // these loads
    __amdgpu_buffer_rsrc_t f_rs;
    unsigned f_base = 0, f_kill = 0, f_add_l = 0, f_add_r = 0;
    auto fetch_setup = [&](int t) {                 // wave-uniform part
        const bool any = t < p.n_sp;
        const int tl = any ? t : 0;
        const int b = tl / tiles_img, rem = tl - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const unsigned char* img = reinterpret_cast<const unsigned char*>(p.x) + (size_t)b * img_bytes;
        f_rs = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, any ? img_bytes : 0u, 0x00020000);
        // which rim flags mean "outside" for this tile, and the wrap of the left / right halo column
        const bool left = tx == 0, right = tx == p.tiles_x - 1;
        f_kill = F_DEAD | (ty == 0 ? F_TOP : 0u) | (ty == p.tiles_y - 1 ? F_BOT : 0u) | (!p.circ && left ? F_LEFT : 0u) |
                 (!p.circ && right ? F_RIGHT : 0u);
        f_add_l = (p.circ && left) ? row_bytes : 0u;
        f_add_r = (p.circ && right) ? 0u - row_bytes : 0u;
        f_base = ((unsigned)(ty * WTH - 1) * (unsigned)p.W + (unsigned)(tx * WTW - 1)) * 128u;      // modulo 2^32; killed where it would be negative
    };
    auto fetch_one = [&](int k) {
        unsigned off = f_base + in_rel[k];
        off += (in_meta[k] & (F_LEFT << 16)) ? f_add_l : 0u;
        off += (in_meta[k] & (F_RIGHT << 16)) ? f_add_r : 0u;
        off = (in_meta[k] & (f_kill << 16)) ? OOR : off;
        rv[k] = __builtin_amdgcn_raw_buffer_load_b128(f_rs, off, 0, 0);
    };
    auto to_lds = [&]() {
#pragma unroll
        for (int k = 0; k < WNIN; ++k)
            if (!(in_meta[k] & (F_DEAD << 16))) a_t[in_meta[k] & 0xffffu] = rv[k];
    };

    auto dma_tile = [&](int t) {                    // the whole input tile of tile t -> the team's image: DPW instructions per wave
        fetch_setup(t);
        dma_setup(t);
        const unsigned lds_img = lds_address(a_t);
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            if (wt * DPW + j < DPIECES) {           // wave-uniform
                const unsigned m = dmeta[j];
                unsigned off = f_base + (m & 0x7ffffff0u);
                off += (m & F_LEFT) ? f_add_l : 0u;
                off += (m & F_RIGHT) ? f_add_r : 0u;
                off = ((m & (f_kill & 15u)) || (m >> 31)) ? OOR : off;
                wres_dma16(d_rs, lds_img + (unsigned)(wt * DPW + j) * 1024u, off);
            }
        }
    };

    // roles inside a team (conv_first2_bf16.hip): wave wt = (row block wt >> 1, column block wt & 1); M-tile 4 rows x 8 columns,
    // lane l31 -> pixel (row l31 >> 3, column l31 & 7)
    const int mrow = wt >> 1, mcol = wt & 1;
    const int a_lane = ((4 * mrow + (l31 >> 3)) * WPITCH + 8 * mcol + (l31 & 7));
    unsigned char* const slab = reinterpret_cast<unsigned char*>(slab_s) + wave * SLAB_BYTES;
    const unsigned relu_floor = p.relu ? 0u : 0x80008000u;      // witw_relu_bf16x2

    f32x16 acc[2];
    // GATE: the four 16-byte gate octets of this lane's four output stores, loaded at the START of the tile's M phase (the 36 MFMA
    // steps hide them; in the V phase a compiler wait for them would also wait for the tile DMA issued behind them, which the
    // compiler cannot see) and waited for with the builtin at the start of the V phase, in front of the DMA
    u32x4 gt[4];
    unsigned gb[4] = {0u, 0u, 0u, 0u};               // GATE = 2: the byte (8 channels) of each of the four stores
    auto bits_mask = [](unsigned byte, int e) -> unsigned {      // dword e of an octet = channels 2e, 2e + 1
        return (((byte >> (2 * e)) & 1u) ? 0x0000ffffu : 0u) | (((byte >> (2 * e + 1)) & 1u) ? 0xffff0000u : 0u);
    };
    auto gate_mask = [](unsigned w) -> unsigned {      // 0xffff per bf16 half that is > 0 (conv3x3_bf16.hip: gate_open)
        const unsigned lo = ((w & 0x7fffu) != 0u && !(w & 0x8000u)) ? 0x0000ffffu : 0u;
        const unsigned hi = ((w & 0x7fff0000u) != 0u && !(w & 0x80000000u)) ? 0xffff0000u : 0u;
        return lo | hi;
    };
    int pb = 0, poy0 = 0, pox0 = 0;                 // the tile whose accumulators are waiting for their epilogue
    bool pvalid = false;
    // ---- epilogue, D[channel][pixel]: lane = pixel l31, registers 4j..4j+3 of accumulator nt = channels nt*32 + 8j + 4hq + {0..3};
    // one row (8 pixels) of the M-tile at a time through the slab: 8 x 128 bytes = 64 lanes x 16 bytes
    auto epilogue = [&]() {
        f32x4 bq[2][4];                             // bq[nt][j][e]: channel cb*64 + nt*32 + 8j + 4hq + e <-> register 4j+e of accumulator nt
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[nt][j] = reinterpret_cast<const f32x4*>(bias_s)[nt * 8 + 2 * j + hq];
        if (WITW_WRES_DIRECT) {
            // straight from the registers: lane (pixel l31, half hq) holds channels 8j + 4hq .. + 3 of every octet j as one 8-byte pair;
            // v_permlane32_swap exchanges the upper half-wave of octet j with the lower half-wave of octet j + 1, after which a lower
            // lane holds all 8 channels of octet j of its pixel and the upper lane of the same pixel all 8 of octet j + 1: one 16-byte
            // store each, 32 contiguous bytes per pixel and instruction, four instructions per wave -- and no LDS traffic under the
            // other team's operand reads (the slab form's writes and read-backs cost 7 % of the kernel in the switch-off builds)
            const int oy = poy0 + 4 * mrow + (l31 >> 3), ox = pox0 + 8 * mcol + (l31 & 7);
            unsigned short* dst = p.y + (((size_t)pb * p.H + oy) * p.W + ox) * p.Cout + cb * 64 + 8 * hq;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    unsigned x[2][2];                   // [octet of the pair][dword]
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const int j = 2 * jp + o;
                        x[o][0] = witw_relu_bf16x2(witw_pack_bf16x2(acc[nt][4 * j] + bq[nt][j][0], acc[nt][4 * j + 1] + bq[nt][j][1]), relu_floor);
                        x[o][1] = witw_relu_bf16x2(witw_pack_bf16x2(acc[nt][4 * j + 2] + bq[nt][j][2], acc[nt][4 * j + 3] + bq[nt][j][3]), relu_floor);
                    }
                    u32x4 v;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(x[0][d], x[1][d], false, false);
                        v[d] = sw[0];
                        v[2 + d] = sw[1];
                    }
                    // plain stores: the four 32-byte pieces of a pixel's 128-byte line must meet in L2 before the line leaves it
                    if (pvalid) *reinterpret_cast<u32x4*>(dst + nt * 32 + 16 * jp) = v;
                }
            return;
        }
        // two rounds of two M-tile rows (16 pixels) through the slab. One wave's LDS operations execute in order, so neither the
        // read-back behind the writes nor the second round's writes behind the first round's reads need a wait of their own: the
        // only waits are the ones in front of the stores (the values' first use). (A first form synchronised four one-row rounds:
        // eight LDS round trips in a row, 2.9 k cycles beside the other team's operand reads.)
        u32x4 v[4];
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if ((l31 >> 4) == rnd && !(WITW_WRES_DIAG & 16)) {
                unsigned char* dst = slab + (l31 & 15) * SLAB_PITCH + 8 * hq;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        u32x2 ob;
                        ob[0] = witw_relu_bf16x2(witw_pack_bf16x2(acc[nt][4 * j] + bq[nt][j][0], acc[nt][4 * j + 1] + bq[nt][j][1]), relu_floor);
                        ob[1] = witw_relu_bf16x2(witw_pack_bf16x2(acc[nt][4 * j + 2] + bq[nt][j][2], acc[nt][4 * j + 3] + bq[nt][j][3]), relu_floor);
                        *reinterpret_cast<u32x2*>(dst + (nt * 32 + 8 * j) * 2) = ob;
                    }
            }
            __builtin_amdgcn_wave_barrier();            // (compiler only: the other lanes' writes stay in front of the reads)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = lane + 64 * i;            // 16 pixels x 8 channel octets
                if (WITW_WRES_DIAG & 16) v[2 * rnd + i] = (u32x4){__builtin_bit_cast(unsigned, acc[0][4 * rnd + i]), __builtin_bit_cast(unsigned, acc[1][4 * rnd + i]), 0u, 0u};
                else v[2 * rnd + i] = *reinterpret_cast<const u32x4*>(slab + (c >> 3) * SLAB_PITCH + (c & 7) * 16);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * (k & 1);
            const int px = c >> 3, c8 = c & 7;          // pixel 0..15 of the round: row px >> 3, column px & 7
            const int oy = poy0 + 4 * mrow + 2 * (k >> 1) + (px >> 3), ox = pox0 + 8 * mcol + (px & 7);
            if (GATE == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] &= gate_mask(gt[k][e]);
            }
            if (GATE == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] &= bits_mask(gb[k], e);
            }
            if (pvalid)
                __builtin_nontemporal_store(v[k], reinterpret_cast<u32x4*>(p.y + (((size_t)pb * p.H + oy) * p.W + ox) * p.Cout + cb * 64 + c8 * 8));
        }
    };

    // the first tile's input goes to LDS here; inside the loop the registers prefetched during an M phase are consumed at ONE place
    // (top of the next V phase), always behind the same sequence of memory operations
    if (WITW_WRES_DMA) {
        dma_tile(tile_of(0));
    } else {
        fetch_setup(tile_of(0));
#pragma unroll
        for (int k = 0; k < WNIN; ++k) fetch_one(k);
        to_lds();
    }
    // everything loaded so far (filter, biases, the first input tile) has landed: said with the builtin, so that the compiler's
    // counter bookkeeping enters the loop clean
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the DMA pieces are outside the compiler's bookkeeping)
    __syncthreads();                                // filter and both teams' first input tiles in LDS
    if (team) __syncthreads();                      // team 1 runs half a tile behind: its V phases meet team 0's M phases

    for (int it = 0; it < n_it; ++it) {
        const bool rec = REC && blockIdx.x == 0 && it == 2 && lane == 0 && wt == 0;
        auto stamp = [&](int k) { if (rec) wres_stamps[team][k] = __builtin_amdgcn_s_memtime(); };
        stamp(0);
        const int t = tile_of(it);
        const bool valid = t < p.n_sp;
        const int tl = valid ? t : 0;
        const int b = tl / tiles_img, rem = tl - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;

        // ================= V phase (the other team is in its M phase) =================
        if (WITW_WRES_PRIO) __builtin_amdgcn_s_setprio(WITW_WRES_PRIO == 2 ? 1 : 0);
        if (it > 0 && !(WITW_WRES_DIAG & 4)) {
            if (WITW_WRES_DMA) {
                // the team's image was last read in the previous M phase, a barrier ago: this tile's pieces go out now and land under
                // the epilogue; the wait counts the epilogue's four stores issued behind them (vector-memory operations retire in
                // issue order), so it does not wait for those stores
                if (GATE) __builtin_amdgcn_s_waitcnt(0x0F70);      // the gate octets (loaded an M phase ago): the compiler's bookkeeping is clean from here
                if (!(WITW_WRES_DIAG & 8)) dma_tile(t);
                stamp(1);
                epilogue();
                if (pvalid) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                // the team's input tile was last read in the previous M phase, a barrier ago; the loads of this tile were issued there
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (!(WITW_WRES_DIAG & 8)) to_lds();
                stamp(1);
                epilogue();
            }
        }
        stamp(2);
        __syncthreads();
        stamp(3);

        // ================= M phase (the other team is in its V phase) =================
        if (WITW_WRES_PRIO) __builtin_amdgcn_s_setprio(WITW_WRES_PRIO == 2 ? 0 : 1);
        // the next tile's loads go out one at a time between the MFMA steps (all six at once queue up behind each other in the
        // texture path and hold the wave at the issue of the last ones)
        asm volatile("" ::: "memory");
        if (GATE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {           // the addresses of the epilogue's stores (below)
                const int c = lane + 64 * (k & 1);
                const int px = c >> 3, c8 = c & 7;
                const int oy = ty * WTH + 4 * mrow + 2 * (k >> 1) + (px >> 3), ox = tx * WTW + 8 * mcol + (px & 7);
                if (GATE == 1) gt[k] = *reinterpret_cast<const u32x4*>(p.gate + (((size_t)b * p.H + oy) * p.W + ox) * p.Cout + cb * 64 + c8 * 8);
                if (GATE == 2) gb[k] = p.gate_bits[(((size_t)b * p.H + oy) * p.W + ox) * (p.Cout >> 3) + cb * 8 + c8];
            }
        }
        if (!WITW_WRES_DMA) fetch_setup(tile_of(it + 1));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        {
            constexpr int PF = WITW_WRES_PF, NB = PF + 1;
            u32x4 fa[NB], fb[NB][2];
            const unsigned abase = lds_address(a_t) + (unsigned)(hq * WPOS + a_lane) * 16u;
            const unsigned dbase = lds_address(a_t);
            const unsigned wbase = lds_address(w_s) + (unsigned)(hq * 64 + l31) * 16u;
            auto a_addr = [&](int step) {
                const int kc = step / 9, tap = step - kc * 9;
                const int kh = tap / 3, kw = tap - kh * 3;
                if (WITW_WRES_DMA) {                 // pixel-major swizzled image: slot 8 * pixel + (group ^ f(row, column))
                    const int R = 4 * mrow + (l31 >> 3) + kh, C = 8 * mcol + (l31 & 7) + kw;
                    const int f = ((C >> 1) & 1) | ((R & 3) << 1);
                    return dbase + (unsigned)((R * WAW + C) * 8 + ((2 * kc + hq) ^ f)) * 16u;
                }
                return abase + (unsigned)(2 * kc * WPOS + kh * WPITCH + kw) * 16u;
            };
            auto w_addr = [&](int step, int nt) { return wbase + (unsigned)(step * 128 + nt * 32) * 16u; };
            int issued = 0;
            auto issue = [&](int step, int which) {
                const int bq = step % NB;
                if (WITW_WRES_DIAG & 2) { ++issued; return; }
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
                if (!(WITW_WRES_DIAG & 3)) lds_wait(issued - (3 * step + 2), fa[bq], fb[bq][0]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[bq][0]), __builtin_bit_cast(bf16x8, fa[bq]), acc[0], 0, 0, 0);
                if (more) {
                    issue(step + PF, 0);
                    issue(step + PF, 1);
                }
                if (!(WITW_WRES_DIAG & 3)) lds_wait(issued - (3 * step + 3), fb[bq][1]);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[bq][1]), __builtin_bit_cast(bf16x8, fa[bq]), acc[1], 0, 0, 0);
                if (more) issue(step + PF, 2);
                if (step % 5 == 2 && step / 5 < WNIN) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!WITW_WRES_DMA) fetch_one(step / 5);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        pb = b; poy0 = ty * WTH; pox0 = tx * WTW; pvalid = valid;
        stamp(4);
        __syncthreads();
        stamp(5);
    }
    if (WITW_WRES_PRIO) __builtin_amdgcn_s_setprio(WITW_WRES_PRIO == 2 ? 1 : 0);
    if (GATE) __builtin_amdgcn_s_waitcnt(0x0F70);
    if (n_it > 0) epilogue();
    if (!team) __syncthreads();                     // team 0 is half a tile ahead: the barrier team 1's last M phase ends on
}

}  // namespace

// Does the weight-resident kernel take this layer? (Cin = 64, stride 1, plain bf16 NHWC forward; enough tiles per workgroup to
// pay for loading the filter block once; WITW_BF_WRES=0 turns it off)
static int g_wres = -1;
static int wres_enabled() {
    if (g_wres < 0) {
        const char* e = getenv("WITW_BF_WRES");
        g_wres = e ? atoi(e) != 0 : 1;
    }
    return g_wres;
}

bool witw_bf16_wres_applies(int B, int H, int W, int Cin, int Cout) {
    if (!wres_enabled() || Cin != 64 || Cout < 64 || (Cout % 64) != 0 || (H % WTH) != 0 || (W % WTW) != 0) return false;
    const long long n_sp = (long long)B * (H / WTH) * (W / WTW);
    return n_sp * (Cout / 64) >= 32 * 256 && n_sp < 0x7fffffffLL;      // 16 tile pairs per workgroup pay for loading the filter block once
}

int witw_bf16_wres_launch(const void* x, const void* wpk, const float* bias, const void* gate, const void* gate_bits, void* y, int B, int H,
                          int W, int Cout, int pad_circular, int relu, void* stream) {
    WresArgs a;
    a.x = (const u32x4*)x; a.wpk = (const u32x4*)wpk; a.bias = bias; a.y = (unsigned short*)y; a.gate = (const unsigned short*)gate;
    a.gate_bits = (const unsigned char*)gate_bits;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout;
    a.tiles_x = W / WTW; a.tiles_y = H / WTH;
    a.n_sp = B * a.tiles_x * a.tiles_y;            // team tiles; a workgroup walks pairs of them
    a.n_cb = Cout / 64;
    a.w_tn = Cout >= 128 ? 128 : 64;
    a.circ = pad_circular; a.relu = relu;
    const int n_cu = witw_cu_count();        // persistent workgroups, one per CU (150 KB of LDS each)
    int q = n_cu / (8 * a.n_cb);
    if (q < 1) q = 1;
    a.q_per_xcd = q;
    const unsigned grid = 8u * (unsigned)(q * a.n_cb);
    const bool rec = !gate && !gate_bits && getenv("WITW_WRES_STAMPS") != nullptr && (a.n_sp + 1) / 2 >= 3 * 8 * q;      // diagnostic, synchronous
    if (rec) {
        hipLaunchKernelGGL((conv3x3_bf16_wres_kernel<true, 0>), dim3(grid), dim3(WRT), 0, (hipStream_t)stream, a);
        (void)hipDeviceSynchronize();
        unsigned long long h[2][8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(wres_stamps), sizeof(h)) == hipSuccess)
            for (int w = 0; w < 2; ++w)
                fprintf(stderr, "conv3x3_bf16_wres team %d (wave %d), third iteration (ticks): V input->LDS %llu, V epilogue %llu, barrier %llu, "
                                "M MFMA loop %llu, barrier %llu, total %llu\n", w, 4 * w, h[w][1] - h[w][0], h[w][2] - h[w][1], h[w][3] - h[w][2],
                        h[w][4] - h[w][3], h[w][5] - h[w][4], h[w][5] - h[w][0]);
    } else if (gate_bits) {
        hipLaunchKernelGGL((conv3x3_bf16_wres_kernel<false, 2>), dim3(grid), dim3(WRT), 0, (hipStream_t)stream, a);
    } else if (gate) {
        hipLaunchKernelGGL((conv3x3_bf16_wres_kernel<false, 1>), dim3(grid), dim3(WRT), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL((conv3x3_bf16_wres_kernel<false, 0>), dim3(grid), dim3(WRT), 0, (hipStream_t)stream, a);
    }
    WITW_CHECK_LAUNCH("conv3x3_bf16_wres");
    witw_note_variant(gate_bits ? "conv3x3_bf16_wres_kernel<gate_bits>" : gate ? "conv3x3_bf16_wres_kernel<gate>" : "conv3x3_bf16_wres_kernel");
    return WITW_OK;
}

extern "C" {

// 1 (default): 64-input-channel plain bf16 forwards with H % WTH == 0, W % WTW == 0, Cout % 64 == 0 and at least 8192 (team tile,
// channel block) units (witw_bf16_wres_applies) run on the weight-resident kernel; 0: on the tiled kernels. enable < 0 only queries. Returns the previous setting.
// The results are bit-identical to the 32x32x16 tiled kernel's.
int witw_conv3x3_bf16_wres(int enable) {
    const int prev = wres_enabled();
    if (enable >= 0) g_wres = enable != 0;
    return prev;
}

}  // extern "C"
