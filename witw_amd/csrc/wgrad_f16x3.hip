// Weight gradient of the 3x3 convolution with fp32-grade products on the fp16 MFMA (the wgrad of the fp16x3 training step,
// conv3x3_f16x3.hip): dW[co][ci][kh][kw] = sum_{b,h,w} dZ[b,h,w,co] * Xpad[b, h*SH+kh-1, w+kw-1, ci] with X and dZ carried as
// fp16 hi + fp16 lo and every product formed as hi*hi + lo*hi + hi*lo (fp32 accumulate).
//
// Reference semantics: autograd through torch.nn.Conv2d in the training loop of model/cvig_fov.py:447-460. Structure =
// wgrad_bf16.hip: the contraction runs over (image, pixel), both operands are re-laid out once per layer into a BATCH-OCTET
// layout so that the 8 k values of an MFMA lane are 8 images at one pixel — here with the two planes:
//   [B/8][H][W][C][2 planes][8 images] fp16   (32 bytes per (pixel, channel): hi octet, lo octet).
// Per pixel pair (p, p+1) and tap three MFMAs: A = [x_hi(p) | x_lo(p)] x B = [dz_hi(p) | dz_hi(p)], the same for p+1, and
// the cross terms of both pixels in one: A = [x_hi(p) | x_hi(p+1)] x B = [dz_lo(p) | dz_lo(p+1)].
// Workgroup = 8 waves, tile 64 (ci) x 64 (co) of all 9 taps; wave (wm, wn, kq): 32 ci x 32 co, pixel pairs {2kq, 2kq+1} of
// the chunk's four (the two kq groups write separate split-K partials). One K chunk = one image octet x one output row x 8
// columns: 3 x 10 halo pixels of X and 8 pixels of dZ, both planes, by LDS-DMA into a [pixel][plane][channel] image
// (conflict-free ds_read_b128), double buffered (2 x 76 KB).
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int WH_P = 8;                 // output columns per chunk
constexpr int WH_XC = WH_P + 2;         // halo columns
constexpr int WH_T = 64;                // ci and co tile of a workgroup
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

struct WgradHxArgs {
    const unsigned short* x;    // [B8][H][W][Cin][2][8]   fp16, batch-octet split layout
    const unsigned short* dz;   // [B8][Ho][Wo][Cout][2][8]
    float* ws;                  // [2*splits][9][Cin][Cout]
    int B8, H, W, Cin, Cout, Ho, Wo;
    int circ;
    int nseg;                   // column segments of WH_P per output row
    int chunks;                 // B8 * Ho * nseg
    int cps;                    // chunks per split
};

template <int SH>
__global__ __launch_bounds__(512) void conv3x3_wgrad_f16x3_kernel(WgradHxArgs p) {
    constexpr int NW = 8;
    constexpr int NXI = 3 * WH_XC * 2;                // X DMA instructions per stage: (halo pixel, plane) -> 64 ci slots
    constexpr int NZI = WH_P * 2;                     // dZ DMA instructions per stage: (pixel, plane) -> 64 co slots
    constexpr int X_S = NXI * 64;                     // 16-B slots
    constexpr int Z_S = NZI * 64;
    constexpr int STAGE_S = X_S + Z_S;
    static_assert(2 * STAGE_S * 16 <= 160 * 1024, "two stages must fit the LDS");
    __shared__ u32x4 stageA[STAGE_S];
    __shared__ u32x4 stageB[STAGE_S];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6) & (NW - 1);
    const int l31 = lane & 31, kg = lane >> 5;
    const int wm = wave_u & 1, wn = (wave_u >> 1) & 1, kq = wave_u >> 2;
    const int ci0 = blockIdx.x * WH_T, co0 = blockIdx.y * WH_T, split = blockIdx.z;
    const int c_begin = split * p.cps;
    const int c_end = min(p.chunks, c_begin + p.cps);

    // one buffer descriptor per image octet (a layer-0 gradient of 128 images is 2.1 GB as a whole: past the 2 GB that
    // keep the out-of-range offset 0x80000000 out of range)
    const size_t x_oct_halves = (size_t)p.H * p.W * p.Cin * 16, z_oct_halves = (size_t)p.Ho * p.Wo * p.Cout * 16;
    // lane -> channel of the tile; its 16 B of one plane sit at channel*32 + plane*16 of the pixel's run
    const unsigned x_lane = (ci0 + lane < p.Cin) ? (unsigned)lane * 32u : OOR;
    const unsigned z_lane = (co0 + lane < p.Cout) ? (unsigned)lane * 32u : OOR;

    // chunk c -> stage s: this wave's share of the DMA instructions; LDS image [pixel][plane][64 channels]
    auto stage = [&](int c, u32x4* s) {
        const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_address(s));
        const int seg = c % p.nseg;
        const int t = c / p.nseg;
        const int h = t % p.Ho, b8 = t / p.Ho;
        const int w0 = seg * WH_P;
        const i32x4 x_rs = raw_rsrc(p.x + (size_t)b8 * x_oct_halves, (unsigned)(x_oct_halves * 2));
        const i32x4 z_rs = raw_rsrc(p.dz + (size_t)b8 * z_oct_halves, (unsigned)(z_oct_halves * 2));
#pragma unroll
        for (int i = 0; i < (NXI + NW - 1) / NW; ++i) {
            const int j = wave_u + NW * i;
            if (NXI % NW == 0 || j < NXI) {
                const int plane = j & 1, px = j >> 1;
                const int r = px / WH_XC, cc = px - r * WH_XC;
                const int gr = h * SH - 1 + r;
                int gc = w0 - 1 + cc;
                bool ok = gr >= 0 && gr < p.H;
                if (p.circ) {               // only columns -1 and W wrap; columns past W pair with zero dZ pixels
                    if (gc < 0) gc += p.W;
                    else if (gc >= p.W) gc -= p.W;
                }
                ok = ok && gc >= 0 && gc < p.W;
                const unsigned soff = ok ? (unsigned)((((size_t)gr * p.W + gc) * p.Cin + ci0) * 32u + plane * 16u) : 0u;
                dma16(x_rs, lds + (unsigned)j * 1024u, ok ? x_lane : OOR, soff);
            }
        }
        static_assert(NZI % NW == 0, "dZ instructions split evenly over the waves");
#pragma unroll
        for (int i = 0; i < NZI / NW; ++i) {
            const int j = wave_u + NW * i;
            const int plane = j & 1, px = j >> 1;
            const int w = w0 + px;
            const bool ok = w < p.Wo;
            const unsigned soff = ok ? (unsigned)((((size_t)h * p.Wo + w) * p.Cout + co0) * 32u + plane * 16u) : 0u;
            dma16(z_rs, lds + (unsigned)(X_S + j * 64) * 16u, ok ? z_lane : OOR, soff);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // slot of (halo pixel q, plane pl, channel c) = (q*2 + pl)*64 + c; dZ: X_S + (pixel*2 + pl)*64 + c
    const int a1 = kg * 64 + wm * 32 + l31;           // MFMA_1 A: plane kg (hi | lo) of one pixel
    const int a2 = kg * 128 + wm * 32 + l31;          // MFMA_2 A: hi plane of pixel p + kg
    const int b1 = X_S + wn * 32 + l31;               // MFMA_1 B: hi plane of one pixel, both lane halves
    const int b2 = X_S + kg * 128 + 64 + wn * 32 + l31;   // MFMA_2 B: lo plane of pixel p + kg

    // the MFMAs of one staged chunk that belong to this wave: pixel pairs 2kq and 2kq+1, 9 taps, 3 MFMAs each
    auto compute = [&](const u32x4* s) {
#pragma unroll 1
        for (int pp = 0; pp < 2; ++pp) {
            const int px = 2 * (2 * kq + pp);                           // first pixel of the pair
            const u32x4 zh0 = s[b1 + (px * 2) * 64];
            const u32x4 zh1 = s[b1 + ((px + 1) * 2) * 64];
            const u32x4 zl = s[b2 + (px * 2) * 64];
            const u32x4* ap = s + px * 128;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int off = ((t / 3) * WH_XC + (t % 3)) * 128;      // halo pixel offset of the tap, in slots
                const u32x4 x0 = ap[a1 + off];
                const u32x4 x1 = ap[a1 + off + 128];
                const u32x4 xh = ap[a2 + off];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x0), __builtin_bit_cast(f16x8, zh0), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x1), __builtin_bit_cast(f16x8, zh1), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xh), __builtin_bit_cast(f16x8, zl), acc[t], 0, 0, 0);
            }
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

    // ---- partial tile -> workspace [split*2 + kq][tap][ci][co]
    float* out = p.ws + (size_t)(split * 2 + kq) * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int co = co0 + wn * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (ci < p.Cin && co < p.Cout) out[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][r];
        }
    }
}

// dW[co][ci][kh][kw] (+)= sum_k ws[k][tap][ci][co]; one thread per (tap, ci, co), co fastest.
__global__ void wgrad_f16x3_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cin, int Cout, int parts,
                                          int accumulate, int cin_real) {
    const size_t n = (size_t)9 * Cin * Cout;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int co = idx % Cout;
    const size_t t = idx / Cout;
    const int ci = t % Cin;
    const int tap = (int)(t / Cin);
    float s = 0.f;
    for (int k = 0; k < parts; ++k) s += ws[(size_t)k * n + idx];
    if (ci >= cin_real) return;
    float* d = dw + ((size_t)co * cin_real + ci) * 9 + tap;
    *d = accumulate ? (*d + s) : s;
}

// split-fp16 NHWC [B][HW][C/8][2][8] -> batch-octet split [ceil(B/8)][HW][C][2][8] (images past B are zeros). One thread
// per (octet, pixel, channel octet): 16 loads of 16 B, two 8x8 transposes in registers, 16 stores of 16 B (512 contiguous bytes).
__global__ void split_to_octet_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int B, size_t HW, int C,
                                      size_t total) {
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int C8 = C >> 3;
    const int c8 = idx % C8;
    const size_t t = idx / C8;
    const size_t pix = t % HW;
    const size_t b8 = t / HW;
    u16x8 in[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const size_t b = b8 * 8 + i;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            if (b < (size_t)B)
                in[pl][i] = *reinterpret_cast<const u16x8*>(x + (((b * HW + pix) * C8 + c8) * 2 + pl) * 8);
            else
                in[pl][i] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    u16x8* out = reinterpret_cast<u16x8*>(y + (((b8 * HW + pix) * C + (size_t)c8 * 8) * 2) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            u16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = in[pl][i][j];
            out[j * 2 + pl] = o;
        }
}

// bias gradient db[co] = sum over images and pixels of dZ (hi + lo) from the split-fp16 NHWC gradient: thread -> (channel
// octet, pixel phase), fixed-order combination through LDS, partials [blocks][C] summed by the finish kernel
__global__ __launch_bounds__(256) void split_channel_sums_kernel(const unsigned short* __restrict__ dz, float* __restrict__ part,
                                                                  size_t npix, int C, int rows_per_block) {
    __shared__ float sh[256][9];
    const int Q = C >> 3;
    const int phases = 256 / Q;
    const int q = threadIdx.x % Q, ph = threadIdx.x / Q;
    const size_t p0 = (size_t)blockIdx.x * rows_per_block;
    const size_t p1 = min(npix, p0 + rows_per_block);
    const _Float16* z = reinterpret_cast<const _Float16*>(dz);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (ph < phases)
        for (size_t px = p0 + ph; px < p1; px += phases) {
            const f16x8 hi = *reinterpret_cast<const f16x8*>(z + ((px * Q + q) * 2) * 8);
            const f16x8 lo = *reinterpret_cast<const f16x8*>(z + ((px * Q + q) * 2 + 1) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)hi[e] + (float)lo[e];
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < Q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = sh[threadIdx.x][e];
            for (int k = 1; k < phases; ++k) t += sh[k * Q + threadIdx.x][e];
            part[(size_t)blockIdx.x * C + 8 * threadIdx.x + e] = t;
        }
    }
}

__global__ void split_bias_finish_kernel(const float* __restrict__ part, float* __restrict__ db, int C, int nparts, int accumulate) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= C) return;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(size_t)k * C + co];
    db[co] = accumulate ? db[co] + s : s;
}

int hx_bias_rows(size_t npix) {         // ~96 partials: the finish kernel walks them serially per channel
    size_t r = (npix + 95) / 96;
    return (int)(r < 32 ? 32 : r);
}

int wgrad_hx_splits(int B8, int Ho, int Wo, int Cin, int Cout) {
    const int tiles = cdiv(Cin, WH_T) * cdiv(Cout, WH_T);
    const int chunks = B8 * Ho * cdiv(Wo, WH_P);
    int splits = cdiv(256, tiles);            // one workgroup per CU
    if (splits > chunks) splits = chunks;
    return splits < 1 ? 1 : splits;
}

}  // namespace

extern "C" {

long long witw_octet_split_elems(int B, int H, int W, int C) { return (long long)cdiv(B, 8) * 8 * H * W * C * 2; }

// x split-fp16 NHWC [B,H,W,C/8,2,8] -> y batch-octet split [ceil(B/8)][H][W][C][2][8]
int witw_split_f16_to_octet(const void* x_split, void* y_oct, int B, int H, int W, int C, void* stream) {
    WITW_CHECK_ARG(x_split && y_oct, "split_f16_to_octet: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "split_f16_to_octet: bad shape (C=%d must be a multiple of 8)", C);
    const size_t total = (size_t)cdiv(B, 8) * H * W * (C / 8);
    hipLaunchKernelGGL(split_to_octet_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x_split, (unsigned short*)y_oct, B, (size_t)H * W, C, total);
    WITW_CHECK_LAUNCH("split_f16_to_octet");
    return WITW_OK;
}

long long witw_conv3x3_wgrad_f16x3_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h) {
    const int Ho = (H + 2 - 3) / stride_h + 1;
    const long long splits = wgrad_hx_splits(cdiv(B, 8), Ho, W, Cin, Cout);
    const size_t npix = (size_t)B * Ho * W;
    const long long bias_parts = (long long)((npix + hx_bias_rows(npix) - 1) / hx_bias_rows(npix));
    return 2 * splits * 9 * Cin * Cout + bias_parts * Cout;
}

// x_oct [B8][H][W][Cin][2][8], dz_oct [B8][Ho][W][Cout][2][8] (witw_split_f16_to_octet), dz_split = the same gradient as
// split-fp16 NHWC (bias gradient; may be NULL with db NULL). dw [Cout][cin_real][3][3] fp32, db [Cout] fp32 or NULL.
int witw_conv3x3_wgrad_f16x3(const void* x_oct, const void* dz_oct, const void* dz_split, float* dw, float* db, float* workspace,
                             int B, int H, int W, int Cin, int cin_real, int Cout, int stride_h, int pad_circular, int accumulate,
                             void* stream) {
    WITW_CHECK_ARG(x_oct && dz_oct && dw && workspace, "conv3x3_wgrad_f16x3: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad_f16x3: bad shape");
    WITW_CHECK_ARG((Cin % 8) == 0 && (Cout % 8) == 0, "conv3x3_wgrad_f16x3: Cin=%d and Cout=%d must be multiples of 8", Cin, Cout);
    WITW_CHECK_ARG(cin_real > 0 && cin_real <= Cin, "conv3x3_wgrad_f16x3: cin_real=%d outside (0,%d]", cin_real, Cin);
    WITW_CHECK_ARG(stride_h == 1 || stride_h == 2, "conv3x3_wgrad_f16x3: stride_h=%d unsupported", stride_h);
    WITW_CHECK_ARG(!db || dz_split, "conv3x3_wgrad_f16x3: the bias gradient needs dz_split");
    WITW_CHECK_ARG(!db || (256 % (Cout / 8)) == 0, "conv3x3_wgrad_f16x3: bias gradient needs Cout/8 to divide 256 (Cout=%d)", Cout);
    const int B8 = cdiv(B, 8);
    const int Ho = (H + 2 - 3) / stride_h + 1;
    WITW_CHECK_ARG((size_t)H * W * Cin * 32 < 0x80000000ull && (size_t)Ho * W * Cout * 32 < 0x80000000ull,
                   "conv3x3_wgrad_f16x3: one image octet of an operand exceeds a buffer descriptor");
    hipStream_t st = (hipStream_t)stream;
    WgradHxArgs a;
    a.x = (const unsigned short*)x_oct; a.dz = (const unsigned short*)dz_oct; a.ws = workspace;
    a.B8 = B8; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Ho = Ho; a.Wo = W;
    a.circ = pad_circular;
    a.nseg = cdiv(a.Wo, WH_P);
    a.chunks = B8 * Ho * a.nseg;
    const int splits = wgrad_hx_splits(B8, Ho, a.Wo, Cin, Cout);
    a.cps = cdiv(a.chunks, splits);
    const dim3 grid(cdiv(Cin, WH_T), cdiv(Cout, WH_T), splits);
    if (stride_h == 2)
        hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<2>), grid, dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<1>), grid, dim3(512), 0, st, a);
    WITW_CHECK_LAUNCH("conv3x3_wgrad_f16x3");
    const size_t n = (size_t)9 * Cin * Cout;
    hipLaunchKernelGGL(wgrad_f16x3_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, dw, Cin, Cout,
                       2 * splits, accumulate, cin_real);
    WITW_CHECK_LAUNCH("wgrad_f16x3_reduce");
    if (db != nullptr) {
        float* part = workspace + (size_t)2 * splits * n;
        const size_t npix = (size_t)B * Ho * W;
        const int rows = hx_bias_rows(npix);
        const int nparts = (int)((npix + rows - 1) / rows);
        hipLaunchKernelGGL(split_channel_sums_kernel, dim3(nparts), dim3(256), 0, st, (const unsigned short*)dz_split, part, npix, Cout,
                           rows);
        hipLaunchKernelGGL(split_bias_finish_kernel, dim3(cdiv(Cout, 256)), dim3(256), 0, st, part, db, Cout, nparts, accumulate);
        WITW_CHECK_LAUNCH("bias_grad_f16x3");
    }
    return WITW_OK;
}

}  // extern "C"
