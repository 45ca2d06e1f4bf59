// First block of cvig_baseline's encoders in one launch (gfx950, fp32 MFMA):
//   x/255 -> -1 + 2x -> Conv2d(C -> 64, k=4, s=2, p=0) -> LeakyReLU(0.2) -> BatchNorm2d (eval: per-channel affine)
// (model/cvig_baseline.py:236-240, 265-268), from the NCHW fp32 image straight to the space-to-depth(2) image the second block's
// Conv2d(k=4, s=2) reads ([B, ceil(vh/2), ceil(vw/2), 4*64], zeros outside the vh x vw valid outputs).
//
// Why its own kernel: as "space-to-depth + 2x2-tap conv" (baseline.hip + conv3x3.hip) the block is a 128 MB re-layout pass (112 us per
// 32 images of 500 x 500) and an MFMA launch whose K is 4 taps x 16 channels, i.e. two chunks between a prologue and an epilogue that
// writes 508 MB: 258 us, 2.0 TB/s, 64 TF/s. Here the raw tile is staged once (3 planes x 18 x 66 floats per 8 x 32 outputs), the
// whole filter (K = 16 C <= 80, 64 channels) lives in the B-operand registers of every wave, and the accumulator layout (lane =
// output channel) stores 128 contiguous bytes per half-wave: the block runs at the speed of its output stream.
//
// K order: k = c*16 + ky*4 + kx, the flat index of torch's [64][C][4][4] filter, so the filter needs no packing. fp32 throughout
// (v_mfma_f32_32x32x2_f32); the sum over k runs in k order inside one accumulator, a different order from the two-chunk form --
// both are plain fp32 sums of 16 C products (parity: the reference's goldens at 1e-4, tests/test_baseline_gpu.py).
#include "common.h"

namespace {

constexpr int FT = 256;                      // 4 waves
constexpr int FTH = 8, FTW = 32;             // output tile: 8 rows x 32 columns; wave w: rows 2w, 2w+1
constexpr int FRH = 2 * FTH + 2, FRW = 2 * FTW + 2;      // raw tile 18 x 66
constexpr int FRP = FRW + 1;                 // LDS row pitch (67): the stride-2 reads of 32 lanes spread over the banks
constexpr int FC_MAX = 5;                    // bands + 2 * orientation

struct First4Args {
    const float* x;          // NCHW fp32 [B,C,H,W]
    const float* w;          // [64][C][4][4]
    const float* bias;       // [64]
    const float* scale;      // [64] or null: y = lrelu(conv + bias) * scale + shift
    const float* shift;
    float* y;                // [B, H2, W2, 256]
    int B, C, H, W, vh, vw, H2, W2;
    int tiles_x, tiles_y, n_tiles;
    int normalize;
    float slope;
};

template <int C>
__global__ __launch_bounds__(FT) void conv4x4s2_first_kernel(First4Args p) {
    __shared__ float raw_s[C * FRH * FRP];
    __shared__ float w_s[64 * (16 * C + 1)];
    constexpr int NSTEP = 8 * C;             // K = 16 C in steps of 2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hq = lane >> 5;

    // ---- once per (persistent) workgroup: filter -> B-operand registers, lane (n = nt*32 + l31, k = 2*step + hq). Through LDS: read
    // straight from the [64][16 C] tensor a wave instruction touches 64 different cache lines (the lanes are 16 C floats apart) -- 16 C
    // such loads per wave were 2/3 of a first version's time -- whereas the staging read is contiguous and the [n][16 C + 1] image is
    // read back without bank conflicts
    constexpr int WP = 16 * C + 1;
    {
        float wv[(64 * 16 * C) / FT];               // 64 * 16 C is a multiple of 256: all loads in flight together
#pragma unroll
        for (int k = 0; k < (64 * 16 * C) / FT; ++k) wv[k] = p.w[tid + k * FT];
#pragma unroll
        for (int k = 0; k < (64 * 16 * C) / FT; ++k) {
            const int i = tid + k * FT;
            w_s[(i / (16 * C)) * WP + i % (16 * C)] = wv[k];
        }
    }
    __syncthreads();
    float wb[2][NSTEP];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) wb[nt][s] = w_s[(nt * 32 + l31) * WP + 2 * s + hq];
    const float bias0 = p.bias[l31], bias1 = p.bias[32 + l31];
    const float sc0 = p.scale ? p.scale[l31] : 1.f, sc1 = p.scale ? p.scale[32 + l31] : 1.f;
    const float sh0 = p.scale ? p.shift[l31] : 0.f, sh1 = p.scale ? p.shift[32 + l31] : 0.f;

    // ---- raw tile of a tile: every load in flight before the first is consumed, a tile AHEAD of the MFMAs; rows / columns past
    // the picture and tiles past the end read zeros (out-of-range offset / empty descriptor: straight-line code)
    const size_t plane = (size_t)p.H * p.W;
    const unsigned img_bytes = (unsigned)(C * plane * 4);      // < 2^31 (checked by the launcher)
    const unsigned out_bytes = (unsigned)p.H2 * (unsigned)p.W2 * 1024u;      // one image's output, < 2^31 as well
    constexpr int NPIX = FRH * FRW, NIT = (NPIX + FT - 1) / FT;
    const int tiles_img = p.tiles_x * p.tiles_y;
    float rv[NIT][C];
    auto fetch = [&](int t) {
        const bool any = t < p.n_tiles;
        const int tt = any ? t : 0;
        const int b = tt / tiles_img, rem = tt - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * C * plane), 0, any ? img_bytes : 0u, 0x00020000);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int s = tid + k * FT;
            const int rr = s / FRW, rc = s - rr * FRW;
            const int gr = 2 * ty * FTH + rr, gc = 2 * tx * FTW + rc;
            const unsigned off = (s < NPIX && gr < p.H && gc < p.W) ? (unsigned)(gr * p.W + gc) * 4u : 0xfffffff0u;
#pragma unroll
            for (int c = 0; c < C; ++c)
                rv[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (unsigned)(c * plane * 4), 0));
        }
    };
    auto consume = [&]() {                          // first use of the prefetched registers: normalise in place
#pragma unroll
        for (int k = 0; k < NIT; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float v = rv[k][c];
                if (p.normalize) {       // x = x / 255.; x = -1. + 2. * x   (model/cvig_baseline.py:265-266)
                    v = v / 255.f;
                    v = -1.f + 2.f * v;
                }
                rv[k][c] = v;
            }
    };
    auto to_lds = [&]() {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int s = tid + k * FT;
            const int rr = s / FRW, rc = s - rr * FRW;
            if (s < NPIX) {
#pragma unroll
                for (int c = 0; c < C; ++c) raw_s[(c * FRH + rr) * FRP + rc] = rv[k][c];
            }
        }
    };

    // One place inside the loop where the prefetched registers are consumed, behind the same sequence of memory operations on
    // every path (conv3x3_bf16_wres.hip), and IN FRONT of the second row's 32 output stores: the wait then counts the loads and the
    // first row's stores (vmcnt(32)); behind all 64 stores it can only be vmcnt(0) -- the stores' round trip, once per tile
    fetch(blockIdx.x);
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): filter, bias, first tile
    consume();
    to_lds();
    __syncthreads();
    for (int t = blockIdx.x; t < p.n_tiles; t += gridDim.x) {
        const int b = t / tiles_img, rem = t - b * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * FTH, ox0 = tx * FTW;
        fetch(t + gridDim.x);

        // ---- per wave: two output rows (M-tiles of 32 columns) x 64 channels
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int orow = 2 * wave + rr;          // output row inside the tile
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            // A operand: lane (m = column l31, k = 2*step + hq) = raw[c][2*orow + ky][2*l31 + kx], k = c*16 + ky*4 + kx
            const float* abase = raw_s + (2 * orow) * FRP + 2 * l31;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int k0 = 2 * s;                // k = k0 + hq: kx = (k0 & 3) + hq (k0 even: no carry into ky)
                const int c = k0 >> 4, ky = (k0 >> 2) & 3, kx0 = k0 & 3;
                const float av = abase[(c * FRH + ky) * FRP + kx0 + hq];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[0][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[1][s], acc[1], 0, 0, 0);
            }
            if (rr == 1) consume();                  // the next tile's pixels have had both MFMA loops to land
            // ---- epilogue: register r of accumulator nt = column (r&3) + 8*(r>>2) + 4*hq, channel nt*32 + l31: a half-wave stores
            // the 32 channels (128 bytes) of one output pixel
            // Buffer stores over image b's output: a pixel outside the padded grid is an out-of-range offset (dropped) -- no branch
            // around the stores, so the compiler can COUNT them in the wait of consume() above
            const int oy = oy0 + orow;
            __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)b * p.H2 * p.W2 * 256), 0, out_bytes, 0x00020000);
            const unsigned row_off = ((unsigned)(oy >> 1) * (unsigned)p.W2 * 256u + (unsigned)(oy & 1) * 128u + (unsigned)l31) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * hq;
                const bool valid = oy < p.vh && ox < p.vw;
                float v0 = acc[0][r] + bias0, v1 = acc[1][r] + bias1;
                v0 = v0 > 0.f ? v0 : v0 * p.slope;
                v1 = v1 > 0.f ? v1 : v1 * p.slope;
                v0 = v0 * sc0 + sh0;
                v1 = v1 * sc1 + sh1;
                const unsigned off = (oy < 2 * p.H2 && ox < 2 * p.W2) ? row_off + ((unsigned)(ox >> 1) * 256u + (unsigned)(ox & 1) * 64u) * 4u : 0xfffffff0u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, valid ? v0 : 0.f), ys, off, 0, 2);        // aux 2: nt
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, valid ? v1 : 0.f), ys, off, 128, 2);
            }
        }
        __syncthreads();                             // every wave has read its rows of the raw tile
        to_lds();                                    // the next tile (zeros past the end)
        __syncthreads();
    }
}

}  // namespace

extern "C" {

// x NCHW fp32 [B,C,H,W] (C <= 5, H, W >= 4) -> y [B, ceil(vh/2), ceil(vw/2), 256] with vh = (H-4)/2+1, vw = (W-4)/2+1:
// y[b, oy/2, ox/2, ((oy&1)*2 + (ox&1))*64 + n] = lrelu(conv4x4s2(f(x))[b,n,oy,ox] + bias[n]) * scale[n] + shift[n] for oy < vh,
// ox < vw, 0 elsewhere; f = (x/255)*2 - 1 when normalize != 0. w: torch layout [64][C][4][4]. scale / shift may both be NULL.
int witw_conv4x4s2_first_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y,
                             int B, int C, int H, int W, int normalize, float lrelu_slope, void* stream) {
    WITW_CHECK_ARG(x && w && bias && y, "conv4x4s2_first: null pointer");
    WITW_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv4x4s2_first: scale and shift come together");
    WITW_CHECK_ARG(B > 0 && C >= 1 && C <= FC_MAX && H >= 4 && W >= 4, "conv4x4s2_first: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    WITW_CHECK_ARG((unsigned long long)C * H * W * 4 < 0x80000000ull && (unsigned long long)H * W * 64 < 0x80000000ull,
                   "conv4x4s2_first: image too large for one buffer descriptor");
    First4Args a;
    a.x = x; a.w = w; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.vh = (H - 4) / 2 + 1; a.vw = (W - 4) / 2 + 1;
    a.H2 = (a.vh + 1) / 2; a.W2 = (a.vw + 1) / 2;
    a.tiles_y = cdiv(2 * a.H2, FTH); a.tiles_x = cdiv(2 * a.W2, FTW);
    a.normalize = normalize; a.slope = lrelu_slope;
    const long long n_tiles = (long long)B * a.tiles_x * a.tiles_y;
    WITW_CHECK_ARG(n_tiles < 0x7fffffffLL, "conv4x4s2_first: too many tiles");
    a.n_tiles = (int)n_tiles;
    const int n_cu = witw_cu_count();        // persistent workgroups, a few per CU (28-35 KB of LDS, ~100 registers each)
    const long long grid = n_tiles < 4LL * n_cu ? n_tiles : 4LL * n_cu;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
    case 1: hipLaunchKernelGGL(conv4x4s2_first_kernel<1>, dim3((unsigned)grid), dim3(FT), 0, st, a); break;
    case 2: hipLaunchKernelGGL(conv4x4s2_first_kernel<2>, dim3((unsigned)grid), dim3(FT), 0, st, a); break;
    case 3: hipLaunchKernelGGL(conv4x4s2_first_kernel<3>, dim3((unsigned)grid), dim3(FT), 0, st, a); break;
    case 4: hipLaunchKernelGGL(conv4x4s2_first_kernel<4>, dim3((unsigned)grid), dim3(FT), 0, st, a); break;
    default: hipLaunchKernelGGL(conv4x4s2_first_kernel<5>, dim3((unsigned)grid), dim3(FT), 0, st, a); break;
    }
    WITW_CHECK_LAUNCH("conv4x4s2_first");
    witw_note_variant("conv4x4s2_first_kernel<%d>", C);
    return WITW_OK;
}

}  // extern "C"
