// 3x3 convolution, bf16 inference, 64 input channels, filter RESIDENT in LDS (gfx950).
//
// Layer 5 of the FOV_DSM trunk (features[5] of VGG16: Conv2d(64 -> 128, 3x3, pad 1) + ReLU on the 64 x 256 map,
// model/cvig_fov.py:261-262) has K = 9 x 64 only: in the tiled kernels (conv3x3_bf16.hip) a workgroup's K loop is 4 chunks long
// and the per-tile prologue (first weight / input stage) + epilogue (accumulators -> LDS -> stores) take 40 % of its time
// (0.38 of the matrix peak against 0.69 on the 512-channel layers). Here a persistent workgroup owns one block of 64 output
// channels, keeps that block's whole filter (4 chunks x 9 taps x 64 channels x 16 bf16 = 73.7 KB) in LDS for all of its tiles
// and only stages the 10 x 34 x 64 input tile of each 8 x 32 output tile -- the layer-2 phase of conv_first2_bf16.hip with the
// input read from HBM instead of being recomputed. The workgroups of the other channel blocks walk the same tiles in the same
// order on the same XCD, so the second read of an input tile is an L2 hit.
//
// Per tile: (1) the input tile, prefetched into registers during the previous tile, -> LDS [channel group of 8][row][pitch 48];
// (2) 4 K chunks x 9 taps of v_mfma_f32_32x32x16_bf16 per wave (wave = row pair x column half: 32 pixels x 64 channels), the
// FILTER as the A operand so that a lane ends up with 4 consecutive channels of one pixel, operands by hand-issued ds_read_b128
// three steps ahead (lds_frag.h); accumulation order chunk-major, tap-minor as in conv3x3_nhwc_bf16_kernel: BIT-IDENTICAL to
// that kernel; (3) bias + ReLU -> bf16 -> a wave-private slab [16 pixels][64 channels] (8-byte writes) -> 16-byte NHWC stores,
// one row of the wave's two at a time.
#include "common.h"
#include "lds_frag.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WRT = 512;                          // 8 waves, two per SIMD
constexpr int WTH = 8, WTW = 32;                  // output tile
constexpr int WAH = WTH + 2, WAW = WTW + 2;       // input tile: 10 x 34 positions
constexpr int WPITCH = 48;                        // LDS row pitch in positions (conv_first2_bf16.hip: the two rows of an M-tile on disjoint banks)
constexpr int WPOS = WAH * WPITCH + 1;            // 481 slots per channel group: the 8 groups of one pixel (8 neighbouring lanes of the staging
                                                  // write) start 4 banks apart -- ds_write_b128 serves 8 contiguous lanes per cycle, bank (a/4) % 32
constexpr int WNIN = (WAH * WAW * 8 + WRT - 1) / WRT;      // 16-byte input chunks per thread and tile (6)
constexpr int SLAB_PITCH = 144;                   // bytes per pixel of the output slab (128 + 16: 8-byte writes of 16 lanes on distinct banks)
constexpr int SLAB_BYTES = 16 * SLAB_PITCH;

struct WresArgs {
    const u32x4* x;           // NHWC bf16 [B,H,W,64] as 16-byte channel groups
    const u32x4* wpk;         // conv3x3_bf16 packing [n_tile][4 chunks][9 taps][2 groups][TN][8 bf16]
    const float* bias;        // [>= Cout]
    unsigned short* y;        // NHWC bf16 [B,H,W,Cout]
    int B, H, W, Cout;
    int tiles_x, tiles_y, n_sp;      // spatial tiles per row / column of an image, in all
    int n_cb;                 // blocks of 64 output channels
    int w_tn;                 // TN of the packing (64 or 128)
    int q_per_xcd;            // tile walkers per XCD and channel block
    int circ, relu;
};

__device__ __forceinline__ void wres_wave_sync() {      // one wave's LDS traffic is processed in issue order: drain the counter, pin the compiler
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ unsigned long long wres_stamps[2][8];     // WITW_WRES_STAMPS=1 diagnostic: phase ticks of waves 0 and 7, third tile of workgroup 0

template <bool REC>
__global__ __launch_bounds__(WRT, 1) void conv3x3_bf16_wres_kernel(WresArgs p) {
    __shared__ u32x4 a_s[8 * WPOS];                 // 61,568 B: input tile
    __shared__ u32x4 w_s[4 * 9 * 2 * 64];           // 73,728 B: this channel block's filter
    __shared__ u32x4 slab_s[8 * SLAB_BYTES / 16];   // 18,432 B: one [16 pixels][64 channels] bf16 slab per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hq = lane >> 5;

    // workgroup -> (XCD, channel block, walker): block i runs on XCD i % 8; the n_cb workgroups of a walker share its tiles
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = slot % p.n_cb, walker = slot / p.n_cb;
    const int t_first = xcd + 8 * walker, t_step = 8 * p.q_per_xcd;

    // ---- once: filter block -> LDS, bias -> registers
    {
        const int per_tile = p.w_tn / 64;           // channel blocks per packed n-tile
        const u32x4* wsrc = p.wpk + (size_t)(cb / per_tile) * (4 * 9 * 2) * p.w_tn + (cb % per_tile) * 64;
        for (int s = tid; s < 4 * 9 * 2 * 64; s += WRT) w_s[s] = wsrc[(size_t)(s >> 6) * p.w_tn + (s & 63)];
    }
    float bv[2][16];                                // register 4j+e of accumulator nt: channel cb*64 + nt*32 + 8j + 4hq + e
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[nt][r] = p.bias[cb * 64 + nt * 32 + 8 * (r >> 2) + 4 * hq + (r & 3)];

    // this thread's input chunks (tile-relative, the same for every tile): 8 lanes = the 128 bytes of one pixel, a wave = 8 neighbouring
    // pixels of a row (whole cache lines, 1 KB contiguous inside a tile row). Per tile only a scalar base is added; the chunks on the
    // tile's rim carry flags, and what a flag means for this tile (padding = out-of-range offset -> the load returns zeros; circular
    // wrap = +-one row length) is decided on scalars: a dozen vector instructions per tile and thread, no branch.
    constexpr unsigned F_TOP = 1, F_BOT = 2, F_LEFT = 4, F_RIGHT = 8, F_DEAD = 16;
    constexpr unsigned OOR = 0xfffffff0u;           // a buffer offset outside any image
    unsigned in_lds[WNIN];                          // slot in a_s
    unsigned in_rel[WNIN];                          // byte offset from the tile's first halo pixel (row -1, column -1)
    unsigned in_flag[WNIN];
#pragma unroll
    for (int k = 0; k < WNIN; ++k) {
        const int s = tid + k * WRT;
        const int g = s & 7, pos = s >> 3;
        const bool live = pos < WAH * WAW;
        const int r = live ? pos / WAW : 0, c = live ? pos - r * WAW : 0;
        in_lds[k] = (unsigned)(g * WPOS + r * WPITCH + c);
        in_rel[k] = ((unsigned)r * (unsigned)p.W + (unsigned)c) * 128u + (unsigned)g * 16u;
        in_flag[k] = (live ? 0u : F_DEAD) | (r == 0 ? F_TOP : 0u) | (r == WAH - 1 ? F_BOT : 0u) | (c == 0 ? F_LEFT : 0u) | (c == WAW - 1 ? F_RIGHT : 0u);
    }
    const int tiles_img = p.tiles_x * p.tiles_y;
    const unsigned img_bytes = (unsigned)p.H * (unsigned)p.W * 128u;      // < 2^31 (checked by the caller)
    const unsigned row_bytes = (unsigned)p.W * 128u;
    u32x4 rv[WNIN];
    // Buffer loads, a tile past the end = empty descriptor: straight-line code, so the wait in front of the LDS write counts exactly
    // these loads (see to_lds below)
    __amdgpu_buffer_rsrc_t f_rs;
    unsigned f_base = 0, f_kill = 0, f_add_l = 0, f_add_r = 0;
    auto fetch_setup = [&](int t) {                 // wave-uniform part
        const bool any = t < p.n_sp;
        const int tt = any ? t : 0;
        const int b = tt / tiles_img, rem = tt - b * tiles_img;
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
        off += (in_flag[k] & F_LEFT) ? f_add_l : 0u;
        off += (in_flag[k] & F_RIGHT) ? f_add_r : 0u;
        off = (in_flag[k] & f_kill) ? OOR : off;
        rv[k] = __builtin_amdgcn_raw_buffer_load_b128(f_rs, off, 0, 0);
    };

    // layer roles (conv_first2_bf16.hip): wave = (row pair, column half); lane l31 -> pixel (row l31 >> 4, column l31 & 15)
    const int prow = wave >> 1, chalf = wave & 1;
    const int a_lane = ((2 * prow + (l31 >> 4)) * WPITCH + 16 * chalf + (l31 & 15));
    unsigned char* slab = reinterpret_cast<unsigned char*>(slab_s) + wave * SLAB_BYTES;
    const unsigned relu_floor = p.relu ? 0u : 0x80008000u;      // witw_relu_bf16x2

    // The first tile's input goes to LDS here, every later one at the bottom of the loop: ONE place inside the loop where the
    // prefetched registers are consumed, always behind the same sequence (6 loads, then the 4 output stores), so that its wait is
    // vmcnt(4..9) -- reached from two paths of different depth it would be vmcnt(0): a store round trip per tile.
    auto to_lds = [&]() {
#pragma unroll
        for (int k = 0; k < WNIN; ++k)
            if (!(in_flag[k] & F_DEAD)) a_s[in_lds[k]] = rv[k];
    };
    fetch_setup(t_first);
#pragma unroll
    for (int k = 0; k < WNIN; ++k) fetch_one(k);
    to_lds();
    __syncthreads();                                // filter and first input tile in LDS
    // everything loaded so far (filter fragments, biases) has landed: said with the builtin, so that the compiler's counter
    // bookkeeping enters the loop clean -- otherwise the first use of such a register INSIDE the loop carries a vmcnt(0) in every
    // iteration, i.e. a wait for the prefetch loads issued just before it
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0), expcnt / lgkmcnt untouched
    int iter = 0;
    for (int t = t_first; t < p.n_sp; t += t_step, ++iter) {
        const bool rec = REC && blockIdx.x == 0 && iter == 2 && lane == 0 && (wave == 0 || wave == 7);
        auto stamp = [&](int k) { if (rec) wres_stamps[wave == 7][k] = __builtin_amdgcn_s_memtime(); };
        stamp(0);
        const int b = t / tiles_img, rem = t - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * WTH, ox0 = tx * WTW;

        // ---- (1) the next tile's loads go out one at a time between the MFMA steps (all six at once queue up behind each other
        // in the texture path and hold the wave at the issue of the last ones)
        asm volatile("" ::: "memory");
        fetch_setup(t + t_step);
        stamp(1);

        // ---- (2) 36 (chunk, tap) steps of 2 MFMAs per wave
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        constexpr int PF = 3, NB = PF + 1;
        u32x4 fa[NB], fb[NB][2];
        const unsigned abase = lds_address(a_s) + (unsigned)(hq * WPOS + a_lane) * 16u;
        const unsigned wbase = lds_address(w_s) + (unsigned)(hq * 64 + l31) * 16u;
        auto a_addr = [&](int step) {
            const int kc = step / 9, tap = step - kc * 9;
            const int kh = tap / 3, kw = tap - kh * 3;
            return abase + (unsigned)(2 * kc * WPOS + kh * WPITCH + kw) * 16u;
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
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[bq][0]), __builtin_bit_cast(bf16x8, fa[bq]), acc[0], 0, 0, 0);
            if (more) {
                issue(step + PF, 0);
                issue(step + PF, 1);
            }
            lds_wait(issued - (3 * step + 3), fb[bq][1]);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[bq][1]), __builtin_bit_cast(bf16x8, fa[bq]), acc[1], 0, 0, 0);
            if (more) issue(step + PF, 2);
            if (step % 5 == 2 && step / 5 < WNIN) {
                __builtin_amdgcn_sched_barrier(0);
                fetch_one(step / 5);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        stamp(2);
        // ---- (3) D[channel][pixel]: lane = pixel l31, registers 4j..4j+3 of accumulator nt = channels nt*32 + 8j + 4hq + {0..3}
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((l31 >> 4) == h) {
                unsigned char* dst = slab + (l31 & 15) * SLAB_PITCH + 8 * hq;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        u32x2 ob;
                        ob[0] = witw_relu_bf16x2(witw_pack_bf16x2(acc[nt][4 * j] + bv[nt][4 * j], acc[nt][4 * j + 1] + bv[nt][4 * j + 1]), relu_floor);
                        ob[1] = witw_relu_bf16x2(witw_pack_bf16x2(acc[nt][4 * j + 2] + bv[nt][4 * j + 2], acc[nt][4 * j + 3] + bv[nt][4 * j + 3]), relu_floor);
                        *reinterpret_cast<u32x2*>(dst + (nt * 32 + 8 * j) * 2) = ob;
                    }
            }
            wres_wave_sync();
            const int oy = oy0 + 2 * prow + h;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = lane + 64 * i;
                const int px = c >> 3, c8 = c & 7;
                const u32x4 v = *reinterpret_cast<const u32x4*>(slab + px * SLAB_PITCH + c8 * 16);
                const int ox = ox0 + 16 * chalf + px;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.y + (((size_t)b * p.H + oy) * p.W + ox) * p.Cout + cb * 64 + c8 * 8));
            }
            wres_wave_sync();
        }
        stamp(3);
        __syncthreads();                            // every wave has left the MFMA loop: a_s may be overwritten
        stamp(4);
        to_lds();
        stamp(5);
        __syncthreads();
        stamp(6);
    }
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
    return n_sp * (Cout / 64) >= 16 * 256 && n_sp < 0x7fffffffLL;
}

int witw_bf16_wres_launch(const void* x, const void* wpk, const float* bias, void* y, int B, int H, int W, int Cout, int pad_circular,
                          int relu, void* stream) {
    WresArgs a;
    a.x = (const u32x4*)x; a.wpk = (const u32x4*)wpk; a.bias = bias; a.y = (unsigned short*)y;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout;
    a.tiles_x = W / WTW; a.tiles_y = H / WTH;
    a.n_sp = B * a.tiles_x * a.tiles_y;
    a.n_cb = Cout / 64;
    a.w_tn = Cout >= 128 ? 128 : 64;
    a.circ = pad_circular; a.relu = relu;
    static int n_cu = 0;        // persistent workgroups, one per CU (150 KB of LDS each)
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    int q = n_cu / (8 * a.n_cb);
    if (q < 1) q = 1;
    a.q_per_xcd = q;
    const unsigned grid = 8u * (unsigned)(q * a.n_cb);
    const bool rec = getenv("WITW_WRES_STAMPS") != nullptr && a.n_sp >= 3 * 8 * q;      // diagnostic, synchronous
    if (rec) {
        hipLaunchKernelGGL(conv3x3_bf16_wres_kernel<true>, dim3(grid), dim3(WRT), 0, (hipStream_t)stream, a);
        (void)hipDeviceSynchronize();
        unsigned long long h[2][8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(wres_stamps), sizeof(h)) == hipSuccess)
            for (int w = 0; w < 2; ++w)
                fprintf(stderr, "conv3x3_bf16_wres wave %d, third tile (ticks): fetch issue %llu, MFMA loop %llu, epilogue %llu, barrier %llu, "
                                "input->LDS %llu, barrier %llu, total %llu\n", w ? 7 : 0, h[w][1] - h[w][0], h[w][2] - h[w][1], h[w][3] - h[w][2],
                        h[w][4] - h[w][3], h[w][5] - h[w][4], h[w][6] - h[w][5], h[w][6] - h[w][0]);
    } else {
        hipLaunchKernelGGL(conv3x3_bf16_wres_kernel<false>, dim3(grid), dim3(WRT), 0, (hipStream_t)stream, a);
    }
    WITW_CHECK_LAUNCH("conv3x3_bf16_wres");
    witw_note_variant("conv3x3_bf16_wres_kernel");
    return WITW_OK;
}

extern "C" {

// 1 (default): 64-input-channel plain bf16 forwards with H % 8 == 0, W % 32 == 0, Cout % 64 == 0 and at least 4096 (tile, channel
// block) units run on the weight-resident kernel; 0: on the tiled kernels. enable < 0 only queries. Returns the previous setting.
// The results are bit-identical to the 32x32x16 tiled kernel's.
int witw_conv3x3_bf16_wres(int enable) {
    const int prev = wres_enabled();
    if (enable >= 0) g_wres = enable != 0;
    return prev;
}

}  // extern "C"
