// Weight gradient of the 3x3 convolution (gfx950, fp32 MFMA) + bias gradient + Adam.
//
// Backward of the trainable FOV_DSM layers (reference: autograd through torch.nn.Conv2d in
// model/cvig_fov.py:447-460; trainable set :275-278): with dZ the gradient at the conv output
// (after the ReLU / Dropout2d gates),
//   dW[co][ci][kh][kw] = sum_{b,h,w} dZ[b,h,w,co] * Xpad[b, h*SH+kh-1, w+kw-1, ci]
// is 9 GEMMs (one per tap) with M = ci, N = co, K = output pixels, sharing the dZ operand. A
// workgroup owns a 64(ci) x 64(co) tile of ALL 9 taps (9 accumulator tiles per wave) and walks a
// contiguous range of K chunks (one chunk = up to 64 pixels of one output row): the 3 x 66 input
// halo rows and the dZ row are staged in LDS in their natural NHWC order ([pixel][channel]), so
// that MFMA lanes (= consecutive channels) read consecutive LDS words. Staging uses buffer
// loads whose descriptor is built per row from scalars: padded rows get a zero-length
// descriptor, padded columns an out-of-range offset (both return 0), so there is no per-load
// VALU work next to the f32 MFMAs. Split-K partials go to a workspace and are summed in a
// fixed order by the reduce kernel (bitwise reproducible), which also writes torch's KCRS layout.
#include "common.h"

namespace {

constexpr int WT = 256;
constexpr int XCOLS = 66;                     // 64 pixels + 2 halo columns
constexpr int X_F = 3 * XCOLS * 64;           // floats of the input tile [3][66][64]
constexpr int DZ_F = 64 * 64;                 // floats of the dZ tile   [64][64]
constexpr int XROW_F4 = XCOLS * 16;           // float4 slots per halo row
constexpr int XLD = (XROW_F4 + WT - 1) / WT;  // loads per thread per halo row (5)
constexpr unsigned OOR = 0xfffffff0u;         // buffer offset that is out of range for any descriptor

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const float* x;    // [B,H,W,Cin]   NHWC
    const float* dz;   // [B,Ho,Wo,Cout] NHWC
    float* ws;         // [splits][9][Cin][Cout]
    float* bias_part;  // nullptr, or [splits][Cout]: partial bias gradients (sum of dZ), written by the first ci tile
    int B, H, W, Cin, Cout, Ho, Wo;
    int SH, circ;
    int nseg;          // column segments of 64 per output row
    int chunks;        // B*Ho*nseg
    int cps;           // chunks per split
};

// TAPS = 9: the full window. TAPS = 4: only the taps (kh, kw) in {1,2}^2 (cvig_baseline's Conv2d(k=4,s=2) as a 2x2
// convolution over the space-to-depth image: the other five taps of its 3x3 form are structurally zero and get no gradient).
// PACK (TAPS = 4, Cin <= 16: cvig_baseline's first block, 12 space-to-depth channels): the M dimension of the MFMA tile
// holds (tap, ci) = 4 x 16 instead of 64 input channels of which 48 would be padding — one accumulator tile per wave
// instead of four, a quarter of the MFMAs; every (tap, ci, co) sum runs over the same pixels in the same order.
template <int TAPS, bool PACK = false>
__global__ __launch_bounds__(WT, 2) void conv3x3_wgrad_kernel(WgradArgs p) {
    static_assert(!PACK || TAPS == 4, "the packed form is the 2x2 sub-window over at most 16 channels");
    constexpr int NACC = PACK ? 1 : TAPS;
    __shared__ float smem[X_F + DZ_F];
    float* x_s = smem;
    float* dz_s = smem + X_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hk = lane >> 5;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
    const int c_begin = split * p.cps;
    const int c_end = min(p.chunks, c_begin + p.cps);

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    constexpr int R0 = (TAPS == 4) ? 1 : 0;      // first halo row that is read
    // bias gradient: the dZ operand every lane reads anyway is summed on the side (one VALU add per 9 MFMAs) by the
    // waves of the first ci tile that own ci rows 0-31 — no separate pass over dZ
    const bool do_bias = p.bias_part != nullptr && blockIdx.x == 0 && (wave & 1) == 0;
    float bsum = 0.f;

    const int mci = (wave & 1) * 32 + l31;    // this lane's ci within the tile (A operand row)
    const int nco = (wave >> 1) * 32 + l31;   // this lane's co within the tile (B operand column)
    const unsigned xrow_bytes = (unsigned)p.W * p.Cin * 4u;
    const unsigned zrow_bytes = (unsigned)p.Wo * p.Cout * 4u;

    int cur_seg = -1;
    unsigned xoff[XLD], zoff[4];
    for (int c = c_begin; c < c_end; ++c) {
        const int seg = c % p.nseg;
        const int bh = c / p.nseg;
        const int h = bh % p.Ho, b = bh / p.Ho;
        const int w0 = seg * 64;
        if (seg != cur_seg) {   // uniform; once per block when the row fits one segment
            cur_seg = seg;
#pragma unroll
            for (int i = 0; i < XLD; ++i) {
                const int s = tid + i * WT;
                const int col = s >> 4, q = s & 15;
                int gc = w0 - 1 + col;
                bool ok = s < XROW_F4 && (ci0 + 4 * q) < p.Cin;
                if (p.circ) {
                    gc %= p.W;
                    if (gc < 0) gc += p.W;
                } else {
                    ok = ok && gc >= 0 && gc < p.W;
                }
                xoff[i] = ok ? ((unsigned)gc * p.Cin + ci0 + 4 * q) * 4u : OOR;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = tid + i * WT;
                const int px = s >> 4, q = s & 15;
                const bool ok = (w0 + px) < p.Wo && (co0 + 4 * q) < p.Cout;
                zoff[i] = ok ? ((unsigned)(w0 + px) * p.Cout + co0 + 4 * q) * 4u : OOR;
            }
        }
        // ---- stage: 3 halo rows of X and one row of dZ
        u32x4 rx[3][XLD], rz[4];
#pragma unroll
        for (int r = R0; r < 3; ++r) {
            const int gr = h * p.SH - 1 + r;
            const bool rok = gr >= 0 && gr < p.H;
            const float* base = p.x + ((size_t)b * p.H + (rok ? gr : 0)) * p.W * p.Cin;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rok ? xrow_bytes : 0u, 0x00020000);
#pragma unroll
            for (int i = 0; i < XLD; ++i) rx[r][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, xoff[i], 0, 0);
        }
        {
            const float* base = p.dz + ((size_t)b * p.Ho + h) * p.Wo * p.Cout;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, zrow_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) rz[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, zoff[i], 0, 0);
        }
        __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
        for (int r = R0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < XLD; ++i) {
                const int s = tid + i * WT;
                if (XLD * WT == XROW_F4 || s < XROW_F4)
                    reinterpret_cast<u32x4*>(x_s)[r * XROW_F4 + s] = rx[r][i];
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) reinterpret_cast<u32x4*>(dz_s)[tid + i * WT] = rz[i];
        __syncthreads();

        // ---- 9 taps x ceil(npix/2) k-steps; lanes 0-31 take pixel 2k, lanes 32-63 pixel 2k+1
        const int npix = min(64, p.Wo - w0);
        const int ksteps = (npix + 1) >> 1;
        const float* bp = dz_s + hk * 64 + nco;
        if constexpr (PACK) {       // A row m = (tap, ci): this wave's taps are (1 + (wave & 1), 1 + (l31 >> 4))
            const float* ap = x_s + ((1 + (wave & 1)) * XCOLS + 1 + (l31 >> 4) + hk) * 64 + (l31 & 15);
#pragma unroll 4
            for (int k = 0; k < ksteps; ++k) {
                const float bv = bp[k * 128];
                if (do_bias) bsum += bv;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k * 128], bv, acc[0], 0, 0, 0);
            }
        } else {
            const float* ap = x_s + (hk + 0) * 64 + mci;
#pragma unroll 2
            for (int k = 0; k < ksteps; ++k) {
                const float bv = bp[k * 128];
                if (do_bias) bsum += bv;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int kh = (TAPS == 4) ? 1 + (t >> 1) : t / 3, kw = (TAPS == 4) ? 1 + (t & 1) : t - (t / 3) * 3;
                    const float av = ap[(kh * XCOLS + kw) * 64 + k * 128];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- partial tile -> workspace [split][tap][ci][co]
    float* out = p.ws + (size_t)split * TAPS * p.Cin * p.Cout;
    const int co = co0 + nco;
    if constexpr (PACK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hk;          // row of this wave's 32: (tap column, ci)
            const int t = (wave & 1) * 2 + (m >> 4), ci = m & 15;
            if (ci < p.Cin && co < p.Cout) out[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[0][r];
        }
    } else {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk;
                if (ci < p.Cin && co < p.Cout) out[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][r];
            }
    }
    if (do_bias) {                       // lanes l and l+32 hold the even / odd pixels of the same co
        bsum += __shfl_xor(bsum, 32, 64);
        if (hk == 0 && co < p.Cout) p.bias_part[(size_t)split * p.Cout + co] = bsum;
    }
}

// dW[co][ci][kh][kw] (+)= sum_split ws[split][tap][ci][co]; one thread per (tap, ci, co), co fastest.
// Threads past the weight elements sum the bias partials: db[co] (+)= sum_split bias_part[split][co].
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cin, int Cout, int splits,
                                    int accumulate, int cin_real, int taps, const float* __restrict__ bias_part,
                                    float* __restrict__ db) {
    const size_t n = (size_t)taps * Cin * Cout;
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
    int tap = (int)(t / Cin);
    if (taps == 4) tap = (1 + (tap >> 1)) * 3 + 1 + (tap & 1);      // 2x2 sub-window -> its place in the 3x3 filter
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * n + idx];
    if (ci >= cin_real) return;   // zero-padded input channels have no weight
    float* d = dw + ((size_t)co * cin_real + ci) * 9 + tap;
    *d = accumulate ? (*d + s) : s;
}

// The same sums when there are many splits and few weights (first layers: one 64 x 64 tile, ~1000 splits — one thread per
// element would walk them serially on a handful of workgroups): a workgroup owns 32 consecutive elements, its 8 groups of
// 32 lanes each add every 8th split, and the group sums are added in group order. The order is fixed (reproducible), but
// differs from wgrad_reduce_kernel's; the launcher chooses by `splits` alone.
constexpr int RW_E = 32, RW_G = 8;
__global__ __launch_bounds__(RW_E * RW_G) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cin,
                                                                         int Cout, int splits, int accumulate, int cin_real, int taps,
                                                                         const float* __restrict__ bias_part, float* __restrict__ db) {
    __shared__ float part[RW_G][RW_E];
    const size_t n = (size_t)taps * Cin * Cout;
    const int e = threadIdx.x % RW_E, g = threadIdx.x / RW_E;
    const size_t idx = (size_t)blockIdx.x * RW_E + e;         // weight elements, then (idx >= n) the bias elements
    const bool is_w = idx < n;
    const size_t co_b = idx - n;
    const bool is_b = !is_w && db != nullptr && co_b < (size_t)Cout;
    const float* src = is_w ? ws + idx : bias_part + co_b;
    const size_t stride = is_w ? n : (size_t)Cout;
    float s = 0.f;
    if (is_w || is_b) {
#pragma unroll 8
        for (int k = g; k < splits; k += RW_G) s += src[(size_t)k * stride];
    }
    part[g][e] = s;
    __syncthreads();
    if (g != 0) return;
#pragma unroll
    for (int j = 1; j < RW_G; ++j) s += part[j][e];
    if (is_b) {
        db[co_b] = accumulate ? db[co_b] + s : s;
        return;
    }
    if (!is_w) return;
    const int co = idx % Cout;
    const size_t t = idx / Cout;
    const int ci = t % Cin;
    int tap = (int)(t / Cin);
    if (taps == 4) tap = (1 + (tap >> 1)) * 3 + 1 + (tap & 1);
    if (ci >= cin_real) return;
    float* d = dw + ((size_t)co * cin_real + ci) * 9 + tap;
    *d = accumulate ? (*d + s) : s;
}

// Backward of the fused MaxPool2d(2,2): dy [B,Hp,Wp,C] (gradient at the pooled output, ReLU gate already
// applied) is routed to the position recorded by the forward (code = dy*2+dx); dx is [B,H,W,C] with H >= 2Hp,
// W >= 2Wp (a dropped odd row/column gets 0). One thread per pooled element and channel quad.
__global__ void maxpool2x2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ code,
                                      float* __restrict__ dx, int Hp, int Wp, int H, int W, int C, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    size_t t = idx / C;
    const int w = t % Wp;
    t /= Wp;
    const int h = t % Hp;
    const size_t b = t / Hp;
    const float g = dy[idx];
    const int k = code[idx];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        dx[((b * H + 2 * h + (q >> 1)) * W + 2 * w + (q & 1)) * C + c] = (q == k) ? g : 0.f;
}

// torch.optim.Adam (no weight decay, no amsgrad) as pinned by the reference (torch==1.8.1,
// model/requirements.txt:1; optimizer built at model/cvig_fov.py:416-418 with lr=1e-5):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.f - b2);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] + (-(lr / bc1)) * (mi / denom);
}

// The same update for up to ADAM_MAX tensors in one launch (an optimizer step over a model is otherwise one launch of a few
// microseconds per parameter tensor): the table travels as the kernel argument, a workgroup finds its tensor from the running
// workgroup counts.
constexpr int ADAM_MAX = 48;
struct AdamTable {
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    unsigned long long n[ADAM_MAX];
    unsigned first_wg[ADAM_MAX + 1];      // workgroups of tensor k: [first_wg[k], first_wg[k+1])
    float bc1[ADAM_MAX], bc2_sqrt[ADAM_MAX];
    int count;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable t, float lr, float b1, float b2, float eps) {
    int k = 0;
    while (k + 1 < t.count && blockIdx.x >= t.first_wg[k + 1]) ++k;
    const size_t i = (size_t)(blockIdx.x - t.first_wg[k]) * 256 + threadIdx.x;
    if (i >= t.n[k]) return;
    float* p = t.p[k];
    const float* g = t.g[k];
    float *m = t.m[k], *v = t.v[k];
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.f - b2);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / t.bc2_sqrt[k] + eps;
    p[i] = p[i] + (-(lr / t.bc1[k])) * (mi / denom);
}

}  // namespace

extern "C" {

// number of K splits the launcher will use and the workspace it needs (floats)
int witw_conv3x3_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout) {
    const int tiles = cdiv(Cin, 64) * cdiv(Cout, 64);
    const int chunks = B * Ho * cdiv(Wo, 64);
    int splits = cdiv(1024, tiles);           // aim at ~1024 workgroups (2 per CU x 256 CUs x 2 rounds)
    if (splits > chunks) splits = chunks;
    if (splits < 1) splits = 1;
    return splits;
}

long long witw_conv3x3_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h) {
    const int Ho = (H + 2 - 3) / stride_h + 1;
    const long long splits = witw_conv3x3_wgrad_splits(B, Ho, W, Cin, Cout);
    return splits * 9 * Cin * Cout + splits * Cout;
}

// x [B,H,W,Cin] NHWC (the conv's input), dz [B,Ho,W,Cout] NHWC (gradient at its output),
// dw [Cout][cin_real][3][3] (torch layout; Cin is the NHWC-padded channel count of x, cin_real <= Cin),
// db [Cout] or NULL. accumulate != 0 adds to dw/db instead of overwriting.
static int wgrad_launch(const float* x, const float* dz, float* dw, float* db, float* workspace, int B, int H, int W, int Cin,
                        int cin_real, int Cout, int stride_h, int pad_circular, int accumulate, int taps, void* stream) {
    WITW_CHECK_ARG(x && dz && dw && workspace, "conv3x3_wgrad: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: bad shape");
    WITW_CHECK_ARG((Cin % 4) == 0 && (Cout % 4) == 0, "conv3x3_wgrad: Cin=%d and Cout=%d must be multiples of 4", Cin, Cout);
    WITW_CHECK_ARG(cin_real > 0 && cin_real <= Cin, "conv3x3_wgrad: cin_real=%d outside (0,%d]", cin_real, Cin);
    WITW_CHECK_ARG(stride_h == 1 || stride_h == 2, "conv3x3_wgrad: stride_h=%d unsupported", stride_h);
    WITW_CHECK_ARG((size_t)W * Cin * 4 < 0xfffffff0ull && (size_t)W * Cout * 4 < 0xfffffff0ull, "conv3x3_wgrad: row too large");
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.x = x; a.dz = dz; a.ws = workspace;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Ho = (H + 2 - 3) / stride_h + 1;
    a.Wo = W;
    a.SH = stride_h; a.circ = pad_circular;
    a.nseg = cdiv(a.Wo, 64);
    a.chunks = B * a.Ho * a.nseg;
    const int splits = witw_conv3x3_wgrad_splits(B, a.Ho, a.Wo, Cin, Cout);
    a.cps = cdiv(a.chunks, splits);
    a.bias_part = db ? workspace + (size_t)splits * 9 * Cin * Cout : nullptr;     // behind the weight partials
    const dim3 grid(cdiv(Cin, 64), cdiv(Cout, 64), splits);
    if (taps == 4) {
        // the five taps outside the 2x2 sub-window get an exact zero gradient
        if (!accumulate && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * cin_real * 9, st) != hipSuccess) {
            witw_set_error("conv3x3_wgrad_taps4: memset failed");
            return WITW_ERR_LAUNCH;
        }
        if (Cin <= 16)
            hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, true>), grid, dim3(WT), 0, st, a);
        else
            hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, false>), grid, dim3(WT), 0, st, a);
    } else {
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<9, false>), grid, dim3(WT), 0, st, a);
    }
    WITW_CHECK_LAUNCH("conv3x3_wgrad");
    const size_t n = (size_t)taps * Cin * Cout;
    if (splits >= 32)
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((n + Cout + RW_E - 1) / RW_E)), dim3(RW_E * RW_G), 0, st,
                           workspace, dw, Cin, Cout, splits, accumulate, cin_real, taps, a.bias_part, db);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + Cout + 255) / 256)), dim3(256), 0, st, workspace, dw, Cin, Cout,
                           splits, accumulate, cin_real, taps, a.bias_part, db);
    WITW_CHECK_LAUNCH("wgrad_reduce");
    return WITW_OK;
}

int witw_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* db, float* workspace, int B, int H, int W, int Cin,
                       int cin_real, int Cout, int stride_h, int pad_circular, int accumulate, void* stream) {
    return wgrad_launch(x, dz, dw, db, workspace, B, H, W, Cin, cin_real, Cout, stride_h, pad_circular, accumulate, 9, stream);
}

// Weight gradient of the 2x2 sub-window form (witw_conv3x3_fwd_taps4 with tap_base = 1, zero padding): only the taps
// (kh, kw) in {1,2}^2 of dw [Cout][cin_real][3][3] are computed, the other five are set to exact zeros. Same workspace
// as witw_conv3x3_wgrad (stride 1).
int witw_conv3x3_wgrad_taps4(const float* x, const float* dz, float* dw, float* db, float* workspace, int B, int H, int W,
                             int Cin, int cin_real, int Cout, int accumulate, void* stream) {
    return wgrad_launch(x, dz, dw, db, workspace, B, H, W, Cin, cin_real, Cout, 1, 0, accumulate, 4, stream);
}

int witw_maxpool2x2_bwd(const float* dy, const unsigned char* code, float* dx, int B, int Hp, int Wp, int H, int W, int C,
                        void* stream) {
    WITW_CHECK_ARG(dy && code && dx, "maxpool2x2_bwd: null pointer");
    WITW_CHECK_ARG(B > 0 && Hp > 0 && Wp > 0 && C > 0 && H >= 2 * Hp && W >= 2 * Wp, "maxpool2x2_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if ((H > 2 * Hp || W > 2 * Wp) &&
        hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * H * W * C, st) != hipSuccess) {   // dropped odd row / column
        witw_set_error("maxpool2x2_bwd: memset failed");
        return WITW_ERR_LAUNCH;
    }
    const size_t total = (size_t)B * Hp * Wp * C;
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, code, dx, Hp, Wp, H, W,
                       C, total);
    WITW_CHECK_LAUNCH("maxpool2x2_bwd");
    return WITW_OK;
}

int witw_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                   float beta2, float eps, int step, void* stream) {
    WITW_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
    WITW_CHECK_ARG(n > 0 && step >= 1, "adam_step: bad n=%lld or step=%d", n, step);
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
    WITW_CHECK_LAUNCH("adam_step");
    return WITW_OK;
}

// witw_adam_step for `count` tensors (host arrays of device pointers, element counts and 1-based step numbers), ceil(count / 48)
// launches; element for element the arithmetic of witw_adam_step.
int witw_adam_step_multi(float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                         const long long* n, const int* step, int count, float lr, float beta1, float beta2, float eps, void* stream) {
    WITW_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n && step && count > 0, "adam_step_multi: null pointer or count=%d", count);
    for (int k = 0; k < count; ++k) {
        WITW_CHECK_ARG(param[k] && grad[k] && exp_avg[k] && exp_avg_sq[k], "adam_step_multi: null pointer in tensor %d", k);
        WITW_CHECK_ARG(n[k] > 0 && step[k] >= 1 && ((n[k] + 255) / 256) < 0x40000000LL, "adam_step_multi: bad n=%lld or step=%d in tensor %d", n[k], step[k], k);
    }
    for (int k0 = 0; k0 < count; k0 += ADAM_MAX) {
        AdamTable t;
        t.count = count - k0 < ADAM_MAX ? count - k0 : ADAM_MAX;
        unsigned long long wgs = 0;
        for (int j = 0; j < t.count; ++j) {
            const int k = k0 + j;
            t.p[j] = param[k]; t.g[j] = grad[k]; t.m[j] = exp_avg[k]; t.v[j] = exp_avg_sq[k];
            t.n[j] = (unsigned long long)n[k];
            t.first_wg[j] = (unsigned)wgs;
            wgs += (unsigned long long)((n[k] + 255) / 256);
            t.bc1[j] = (float)(1.0 - pow((double)beta1, step[k]));
            t.bc2_sqrt[j] = (float)sqrt(1.0 - pow((double)beta2, step[k]));
        }
        WITW_CHECK_ARG(wgs < 0x7fffffffULL, "adam_step_multi: too many elements for one launch");
        t.first_wg[t.count] = (unsigned)wgs;
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, t, lr, beta1, beta2, eps);
        WITW_CHECK_LAUNCH("adam_step_multi");
    }
    return WITW_OK;
}

}  // extern "C"
