// cvig_baseline pieces that are not the conv itself (gfx950; small memory-bound kernels).
//
// The reference's SurfaceEncoder / OverheadEncoder (model/cvig_baseline.py:228-283) stack
// 7 x [Conv2d(k=4,s=2,p=0) -> LeakyReLU(0.2) -> BatchNorm2d]. A 4x4 stride-2 convolution is a
// 2x2 stride-1 convolution over the space-to-depth(2) image (channels (dy,dx,c)), which the
// NHWC MFMA conv kernel runs as a 3x3 filter whose first row/column of taps is zero; LeakyReLU
// and the (eval-mode) BatchNorm affine sit in that kernel's epilogue. Here: the space-to-depth
// re-layout (with the in-model x/255, -1+2x of :265-266 fused for the first layer), the GeM-like
// pooling (:276-282), the final f/sqrt(|f|) (:284), squared-Euclidean distance matrix and the
// exhaustive minibatch triplet loss (:286-315).
#include "common.h"

namespace {

// x: NHWC [B,Hp,Wp,C] (or NCHW [B,C,Hp,Wp] if in_nchw) of which rows < H, cols < W are valid.
// y: NHWC [B,ceil(H/2),ceil(W/2),Cp], y[b,h2,w2,(dy*2+dx)*C+c] = f(x[b,2h2+dy,2w2+dx,c]); zero elsewhere.
__global__ void space_to_depth2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int Hp, int Wp, int H, int W,
                                       int C, int Cp, int in_nchw, int normalize, size_t total,
                                       const float* __restrict__ scale, const float* __restrict__ shift) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const int cc = idx % Cp;
    size_t t = idx / Cp;
    const int w2 = t % W2;
    t /= W2;
    const int h2 = t % H2;
    const int b = (int)(t / H2);
    float v = 0.f;
    if (cc < 4 * C) {
        const int q = cc / C, c = cc - q * C;
        const int h = 2 * h2 + (q >> 1), w = 2 * w2 + (q & 1);
        if (h < H && w < W) {
            v = in_nchw ? x[(((size_t)b * C + c) * Hp + h) * Wp + w] : x[(((size_t)b * Hp + h) * Wp + w) * C + c];
            if (normalize) {       // x = x / 255.; x = -1. + 2. * x   (model/cvig_baseline.py:265-266)
                v = v / 255.f;
                v = -1.f + 2.f * v;
            }
            if (scale != nullptr) v = v * scale[c] + shift[c];     // train-mode BatchNorm of the producer layer
        }
    }
    y[idx] = v;
}

// The same for an NHWC source with C % 4 == 0 and Cp == 4C (every block but the first): one thread per channel quad, 16-byte
// loads and stores, the same arithmetic per element.
__global__ void space_to_depth2_quad_kernel(const float* __restrict__ x, float* __restrict__ y, int Hp, int Wp, int H, int W, int C,
                                            size_t total4, const float* __restrict__ scale, const float* __restrict__ shift) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1, C4 = C >> 2;
    const int c4 = idx % C4;
    size_t t = idx / C4;
    const int q = t & 3;
    t >>= 2;
    const int w2 = t % W2;
    t /= W2;
    const int h2 = t % H2;
    const size_t b = t / H2;
    const int h = 2 * h2 + (q >> 1), w = 2 * w2 + (q & 1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (h < H && w < W) {
        v = *reinterpret_cast<const f32x4*>(x + ((b * Hp + h) * Wp + w) * C + 4 * c4);
        if (scale != nullptr) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * c4), sf = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] * sc[j] + sf[j];
        }
    }
    reinterpret_cast<f32x4*>(y)[idx] = v;
}

// Mosaic space-to-depth: x NHWC [B,H,W,C] (all valid) -> y [ceil(B/g^2), g*H2, g*W2, 4C], H2 = ceil(H/2): image b sits in cell
// (b % g^2) of mosaic image b / g^2, cells row-major, each the space-to-depth(2) image of x[b]; zeros where no source exists.
// A k=4,s=2 layer over an H2 x W2 s2d map has (H2-1) x (W2-1) valid outputs, so no valid output's 2x2 window crosses a cell:
// the deep cvig_baseline layers (8x8 and 4x4 maps) run as g = 2 / g = 4 mosaics that fill the conv kernel's 16x16 tile.
__global__ void s2d_mosaic_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int g, size_t total4) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const int C4 = C >> 2;
    const int c4 = idx % C4;
    size_t t = idx / C4;
    const int q = t & 3;
    t >>= 2;
    const int X = t % (g * W2);
    t /= (g * W2);
    const int Y = t % (g * H2);
    const int bm = (int)(t / (g * H2));
    const int cy = Y / H2, y2 = Y - cy * H2, cx = X / W2, x2 = X - cx * W2;
    const int b = (bm * g + cy) * g + cx;
    const int h = 2 * y2 + (q >> 1), w = 2 * x2 + (q & 1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (b < B && h < H && w < W) v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + h) * W + w) * C + 4 * c4);
    reinterpret_cast<f32x4*>(y)[idx] = v;
}

// Epilogue of a split-K taps4 convolution run over a g x g mosaic: ws [S][Bm, g*h, g*w, C] raw partial sums ->
// y [B, vh, vw, C] (valid outputs of every image, compact), y = affine(lrelu(sum_s ws[s] + bias)); the S partials are added in
// slice order (deterministic).
__global__ void splitk_finish_kernel(const float* __restrict__ ws, int S, size_t stride, const float* __restrict__ bias, int act,
                                     float slope, const float* __restrict__ post_scale, const float* __restrict__ post_shift,
                                     float* __restrict__ y, int B, int g, int h, int w, int vh, int vw, int C, size_t total4) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const int C4 = C >> 2;
    const int c4 = idx % C4;
    size_t t = idx / C4;
    const int xx = t % vw;
    t /= vw;
    const int yy = t % vh;
    const int b = (int)(t / vh);
    const int bm = b / (g * g), cell = b - bm * g * g, cy = cell / g, cx = cell - cy * g;
    const size_t o = ((((size_t)bm * g * h + cy * h + yy) * (g * w)) + cx * w + xx) * C + 4 * c4;
    f32x4 a = *reinterpret_cast<const f32x4*>(ws + o);
    for (int s = 1; s < S; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)s * stride + o);
        a += v;
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * c4);
    f32x4 ps = {1.f, 1.f, 1.f, 1.f}, pt = {0.f, 0.f, 0.f, 0.f};
    if (post_scale != nullptr) {
        ps = *reinterpret_cast<const f32x4*>(post_scale + 4 * c4);
        pt = *reinterpret_cast<const f32x4*>(post_shift + 4 * c4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = a[e] + bv[e];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = v > 0.f ? v : v * slope;
        a[e] = v * ps[e] + pt[e];
    }
    reinterpret_cast<f32x4*>(y)[idx] = a;
}

// f[b, col0 + c] = (mean_{h<H,w<W} relu(x[b,h,w,c])^p)^(1/p); x NHWC [B,Hp,Wp,C]; one thread per (b,c).
constexpr int GEM_PH = 8;      // pixel phases per channel
__global__ __launch_bounds__(64 * GEM_PH) void gem_pool_kernel(const float* __restrict__ x, float* __restrict__ f, int B, int Hp, int Wp, int H,
                                                               int W, int C, int ldf, int col0, float p, const float* __restrict__ scale,
                                                               const float* __restrict__ shift) {
    // one workgroup per (image, 64 channels): a thread sums every GEM_PH-th pixel of its channel (a wave reads 256 contiguous bytes per
    // pixel), the phase sums are added in phase order through LDS. (One thread per (image, channel) walked the 13 x 13 map of block 5
    // serially behind a powf each: 107-124 us per call on 64 workgroups.)
    __shared__ float part[GEM_PH][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl, b = blockIdx.x;
    float s = 0.f;
    if (c < C) {
        const float sc = scale != nullptr ? scale[c] : 1.f, sh = scale != nullptr ? shift[c] : 0.f;
        const int n = H * W;
        for (int q = ph; q < n; q += GEM_PH) {
            const int h = q / W, w = q - h * W;
            float v = x[(((size_t)b * Hp + h) * Wp + w) * C + c];
            if (scale != nullptr) v = v * sc + sh;
            s += powf(fmaxf(v, 0.f), p);
        }
    }
    part[ph][cl] = s;
    __syncthreads();
    if (ph != 0 || c >= C) return;
#pragma unroll
    for (int k = 1; k < GEM_PH; ++k) s += part[k][cl];
    f[(size_t)b * ldf + col0 + c] = powf(s / (float)(H * W), 1.f / p);
}

// f[b,:] /= |f[b,:]|^0.5   (model/cvig_baseline.py:284); one block per row.
__global__ __launch_bounds__(256) void embed_normalize_kernel(float* __restrict__ f, int n) {
    __shared__ float part[4];
    float* row = f + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += row[i] * row[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nrm = sqrtf((part[0] + part[1]) + (part[2] + part[3]));
    const float den = powf(nrm, 0.5f);
    for (int i = threadIdx.x; i < n; i += 256) row[i] = row[i] / den;
}

// D[i][j] = sum_k (a[i][k] - b[j][k])^2 (optionally its square root); block = one i x 256 j.
__global__ __launch_bounds__(256) void pairwise_sqdist_kernel(const float* __restrict__ a, const float* __restrict__ bm,
                                                               float* __restrict__ D, int Na, int Nb, int n, int take_sqrt) {
    extern __shared__ float arow[];
    const int i = blockIdx.y;
    for (int k = threadIdx.x; k < n; k += 256) arow[k] = a[(size_t)i * n + k];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Nb) return;
    const float* br = bm + (size_t)j * n;
    float s = 0.f;
    for (int k = 0; k < n; ++k) {
        const float d = br[k] - arow[k];
        s += d * d;
    }
    D[(size_t)i * Nb + j] = take_sqrt ? sqrtf(s) : s;
}

__device__ __forceinline__ float trip(float x, int soft, float alpha, float margin) {
    return soft ? logf(1.f + expf(alpha * x)) : fmaxf(x + margin, 0.f);
}

// part[i] = sum_{j != i} l(D_ii - D_ij) + l(D_ii - D_ji); D[i][j] = |e1_i - e2_j|^2
__global__ __launch_bounds__(256) void exhaustive_partials_kernel(const float* __restrict__ D, float* __restrict__ part, int B,
                                                                   int soft, float alpha, float margin) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const float dii = D[(size_t)i * B + i];
    float s = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        if (j == i) continue;
        s += trip(dii - D[(size_t)i * B + j], soft, alpha, margin);
        s += trip(dii - D[(size_t)j * B + i], soft, alpha, margin);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[i] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int n,
                                                          float norm) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / norm;
}


// ---------------------------------------------------------------------------------------------------------------
// Training-mode pieces (model/cvig_baseline.py:267-284 under .train(), autograd of :286-315): batch statistics of
// BatchNorm2d, its backward fused with the LeakyReLU backward, GeM / normalisation / loss backward, depth-to-space.

// per-channel partial sums over the valid region: part[blk][0][c] = sum a, part[blk][1][c] = sum a^2
// (bwd form: a2 != nullptr -> sum g and sum g*xhat with xhat = (a - mean[c]) * invstd[c], g = second tensor)
__global__ __launch_bounds__(256) void channel_sums_kernel(const float* __restrict__ a, const float* __restrict__ g,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ part, int Hp, int Wp, int H, int W, int C,
                                                            size_t npix, int rows_per_block, int g_cp) {
    __shared__ float sh[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const size_t p0 = (size_t)blockIdx.y * rows_per_block, p1 = min(npix, p0 + (size_t)rows_per_block);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float mu = g ? mean[c] : 0.f, is = g ? invstd[c] : 0.f;
        for (size_t px = p0 + ph; px < p1; px += 4) {      // px enumerates VALID pixels (b, h<H, w<W)
            const size_t w = px % W, t = px / W, h = t % H, b = t / H;
            const size_t off = ((b * Hp + h) * Wp + w) * C + c;
            const float v = a[off];
            if (g) {
                // g_cp > 0: the gradient is still in the space-to-depth layout the data-gradient conv wrote ([B,ceil(H/2),ceil(W/2),g_cp],
                // channel ((h&1)*2+(w&1))*C + c): read in place, no depth-to-space pass
                const float gv = g_cp ? g[((b * ((H + 1) >> 1) + (h >> 1)) * ((W + 1) >> 1) + (w >> 1)) * g_cp + ((h & 1) * 2 + (w & 1)) * C + c] : g[off];
                s0 += gv;
                s1 += gv * ((v - mu) * is);
            } else {
                s0 += v;
                s1 += v * v;
            }
        }
    }
    sh[0][ph][threadIdx.x & 63] = s0;
    sh[1][ph][threadIdx.x & 63] = s1;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        const int l = threadIdx.x;
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = (sh[0][0][l] + sh[0][1][l]) + (sh[0][2][l] + sh[0][3][l]);
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = (sh[1][0][l] + sh[1][1][l]) + (sh[1][2][l] + sh[1][3][l]);
    }
}

// The same sums for channel counts with C % 4 == 0 and (C/4) | 256 (every layer of the reference's encoders): a block
// owns whole image rows (b, h) so the inner loop has no index arithmetic, a thread owns one channel quad (16-byte loads,
// a wave reads 1 KB contiguous) and every (256 / (C/4))-th pixel of the row; fixed-order combination through LDS.
__global__ __launch_bounds__(256) void channel_sums_rows_kernel(const float* __restrict__ a, const float* __restrict__ g,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* __restrict__ part, int Hp, int Wp, int H, int W, int C,
                                                                 int nrows, int rows_per_block, int g_cp) {
    __shared__ f32x4 sh[2][256];
    const int Q = C >> 2, phases = 256 / Q;
    const int q = threadIdx.x % Q, ph = threadIdx.x / Q;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
    if (g) {
        mu = *reinterpret_cast<const f32x4*>(mean + 4 * q);
        is = *reinterpret_cast<const f32x4*>(invstd + 4 * q);
    }
    for (int r = r0; r < r1; ++r) {
        const int b = r / H, h = r - b * H;
        const size_t row = ((size_t)b * Hp + h) * Wp * C + 4 * q;
        const size_t grow = ((size_t)b * ((H + 1) >> 1) + (h >> 1)) * ((W + 1) >> 1) * g_cp + (h & 1) * 2 * C + 4 * q;      // s2d layout (g_cp > 0)
        for (int w = ph; w < W; w += phases) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a + row + (size_t)w * C);
            if (g) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g_cp ? g + grow + (size_t)(w >> 1) * g_cp + (w & 1) * C : g + row + (size_t)w * C);
                s0 += gv;
                s1 += gv * ((v - mu) * is);
            } else {
                s0 += v;
                s1 += v * v;
            }
        }
    }
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < Q) {
        f32x4 t0 = sh[0][threadIdx.x], t1 = sh[1][threadIdx.x];
        for (int k = 1; k < phases; ++k) {
            t0 += sh[0][k * Q + threadIdx.x];
            t1 += sh[1][k * Q + threadIdx.x];
        }
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + 4 * threadIdx.x) = t0;
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + 4 * threadIdx.x) = t1;
    }
}

// Second stage of the per-channel sums: part [nparts][2][C] -> two sums per channel. A workgroup takes 32 channels; its 32 groups of 32
// lanes each add every 32nd partial row (loads of a group: 128 contiguous bytes, 8 in flight), then the group sums are added in a
// fixed order through LDS -- bit-reproducible, and 32x fewer dependent loads per thread than one thread per channel (the stage took
// 157 us of pure load latency per call at 1024 partial rows; 25-40 us with 8 groups).
constexpr int BNF_CH = 32, BNF_LANES = 32;
__device__ __forceinline__ bool bn_finish_sums(const float* __restrict__ part, int nparts, int C, int& c, float& s0, float& s1) {
    __shared__ float red[2][BNF_LANES][BNF_CH];
    const int cl = threadIdx.x % BNF_CH, pl = threadIdx.x / BNF_CH;
    c = blockIdx.x * BNF_CH + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < C)
#pragma unroll 8
        for (int k = pl; k < nparts; k += BNF_LANES) {
            a0 += part[((size_t)k * 2 + 0) * C + c];
            a1 += part[((size_t)k * 2 + 1) * C + c];
        }
    red[0][pl][cl] = a0;
    red[1][pl][cl] = a1;
    __syncthreads();
    if (pl != 0 || c >= C) return false;
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int j = 0; j < BNF_LANES; ++j) { s0 += red[0][j][cl]; s1 += red[1][j][cl]; }
    return true;
}

// BatchNorm2d training statistics: mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale, and the
// running-stat update running = (1-m)*running + m*stat (unbiased variance), torch semantics.
__global__ __launch_bounds__(BNF_CH * BNF_LANES) void bn_stats_finish_kernel(const float* __restrict__ part, int nparts, int C, float n, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, float momentum, float* __restrict__ mean,
                                       float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
                                       float* __restrict__ running_mean, float* __restrict__ running_var) {
    int c;
    float s0, s1;
    if (!bn_finish_sums(part, nparts, C, c, s0, s1)) return;
    const float mu = s0 / n;
    const float var = fmaxf(s1 / n - mu * mu, 0.f);
    const float is = 1.f / sqrtf(var + eps);
    mean[c] = mu;
    invstd[c] = is;
    scale[c] = gamma[c] * is;
    shift[c] = beta[c] - mu * gamma[c] * is;
    if (running_mean != nullptr) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (var * n / (n - 1.f));
    }
}

// sums[0][c] = sum dy, sums[1][c] = sum dy*xhat  (+ writes dgamma = sums[1], dbeta = sums[0])
__global__ __launch_bounds__(BNF_CH * BNF_LANES) void bn_bwd_finish_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ sums,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    int c;
    float s0, s1;
    if (!bn_finish_sums(part, nparts, C, c, s0, s1)) return;
    sums[c] = s0;
    sums[C + c] = s1;
    dbeta[c] = s0;
    dgamma[c] = s1;
}

// dz = lrelu'(a) * gamma*invstd * (dy - sum_dy/n - xhat*sum_dyxhat/n) on the valid region, 0 elsewhere.
__global__ void bn_lrelu_bwd_apply_kernel(const float* __restrict__ a, const float* __restrict__ dy, float* __restrict__ dz,
                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                          const float* __restrict__ gamma, const float* __restrict__ sums, int Hp, int Wp, int H,
                                          int W, int C, float n, float slope, size_t total, int g_cp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    size_t t = idx / C;
    const int w = t % Wp;
    t /= Wp;
    const int h = t % Hp;
    float out = 0.f;
    if (h < H && w < W) {
        const float av = a[idx];
        const float xh = (av - mean[c]) * invstd[c];
        const size_t b = t / Hp;
        const float gy = g_cp ? dy[((b * ((H + 1) >> 1) + (h >> 1)) * ((W + 1) >> 1) + (w >> 1)) * g_cp + ((h & 1) * 2 + (w & 1)) * C + c] : dy[idx];
        const float da = gamma[c] * invstd[c] * (gy - sums[c] / n - xh * sums[C + c] / n);
        out = av > 0.f ? da : da * slope;
    }
    dz[idx] = out;
}

// The same for C % 4 == 0 (and g_cp % 4 == 0): one thread per channel quad, the same arithmetic per element.
__global__ void bn_lrelu_bwd_apply_quad_kernel(const float* __restrict__ a, const float* __restrict__ dy, float* __restrict__ dz,
                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                               const float* __restrict__ gamma, const float* __restrict__ sums, int Hp, int Wp, int H,
                                               int W, int C, float n, float slope, size_t total4, int g_cp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const int C4 = C >> 2;
    const int c = 4 * (int)(idx % C4);
    size_t t = idx / C4;
    const int w = t % Wp;
    t /= Wp;
    const int h = t % Hp;
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (h < H && w < W) {
        const size_t b = t / Hp;
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[idx];
        const f32x4 gy = *reinterpret_cast<const f32x4*>(
            g_cp ? dy + ((b * ((H + 1) >> 1) + (h >> 1)) * ((W + 1) >> 1) + (w >> 1)) * g_cp + ((h & 1) * 2 + (w & 1)) * C + c : dy + 4 * idx);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c), s1 = *reinterpret_cast<const f32x4*>(sums + C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (av[j] - mu[j]) * is[j];
            const float da = ga[j] * is[j] * (gy[j] - s0[j] / n - xh * s1[j] / n);
            out[j] = av[j] > 0.f ? da : da * slope;
        }
    }
    reinterpret_cast<f32x4*>(dz)[idx] = out;
}

// depth-to-space(2) of the s2d-layout gradient: dx[b,h,w,c] = g[b,h/2,w/2,((h&1)*2+(w&1))*C+c] (+ add[b,h,w,c]) on
// the valid region of a [B,Hp,Wp,C] tensor, 0 elsewhere.
__global__ void depth_to_space2_kernel(const float* __restrict__ g, const float* __restrict__ add, float* __restrict__ dx, int Hp,
                                       int Wp, int H, int W, int C, int Cp, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    size_t t = idx / C;
    const int w = t % Wp;
    t /= Wp;
    const int h = t % Hp;
    const size_t b = t / Hp;
    float v = 0.f;
    if (h < H && w < W) {
        const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
        v = g[((b * H2 + (h >> 1)) * W2 + (w >> 1)) * Cp + ((h & 1) * 2 + (w & 1)) * C + c];
        if (add != nullptr) v += add[idx];
    }
    dx[idx] = v;
}

// GeM backward: dy[b,h,w,c] (+)= df[b,col0+c] * m^(1/p-1) * relu(y)^(p-1) / N, y = a*scale+shift, m = f^p.
__global__ void gem_pool_bwd_kernel(const float* __restrict__ a, const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ f, const float* __restrict__ df, float* __restrict__ dy, int Hp,
                                    int Wp, int H, int W, int C, int ldf, int col0, float p, int accumulate, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    size_t t = idx / C;
    const int w = t % Wp;
    t /= Wp;
    const int h = t % Hp;
    const size_t b = t / Hp;
    float out = 0.f;
    if (h < H && w < W) {
        const float y = fmaxf(a[idx] * scale[c] + shift[c], 0.f);
        const float fv = f[b * ldf + col0 + c];              // = m^(1/p)
        if (y > 0.f && fv > 0.f) out = df[b * ldf + col0 + c] * powf(fv, 1.f - p) * powf(y, p - 1.f) / (float)(H * W);
    }
    dy[idx] = accumulate ? dy[idx] + out : out;
}

// f = g / |g|^(1/2)  ->  dg = df/|g|^(1/2) - 0.5 * (g.df) * g / |g|^(5/2); one block per row. f and df given, g = f*|g|^(1/2)
// is recovered from the saved norm.
__global__ __launch_bounds__(256) void embed_normalize_bwd_kernel(const float* __restrict__ g, const float* __restrict__ df,
                                                                   float* __restrict__ dg, int n) {
    __shared__ float part[2][4];
    const float* gr = g + (size_t)blockIdx.x * n;
    const float* dr = df + (size_t)blockIdx.x * n;
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        s0 += gr[i] * gr[i];
        s1 += gr[i] * dr[i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = s0;
        part[1][threadIdx.x >> 6] = s1;
    }
    __syncthreads();
    const float nn = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);     // |g|^2
    const float gd = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    const float nrm = sqrtf(nn);
    const float c0 = 1.f / sqrtf(nrm), c1 = 0.5f * gd / (nrm * nrm * sqrtf(nrm));
    for (int i = threadIdx.x; i < n; i += 256) dg[(size_t)blockIdx.x * n + i] = dr[i] * c0 - c1 * gr[i];
}

__device__ __forceinline__ float trip_d(float x, int soft, float alpha, float margin) {
    return soft ? alpha / (1.f + expf(-alpha * x)) : ((x + margin > 0.f) ? 1.f : 0.f);
}

// G = dL/dD for the exhaustive loss; one block per row a (diagonal entry gathers the row and column terms).
__global__ __launch_bounds__(256) void exhaustive_bwd_kernel(const float* __restrict__ D, const float* __restrict__ gloss,
                                                              float* __restrict__ G, int B, int soft, float alpha, float margin) {
    __shared__ float sh[4];
    const int a = blockIdx.x;
    const float daa = D[(size_t)a * B + a];
    const float sc = gloss[0] / (2.f * B * (B - 1));
    float diag = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        if (j == a) continue;
        const float dab = D[(size_t)a * B + j];
        const float t1 = trip_d(daa - dab, soft, alpha, margin);                       // anchor a in embed1, negative j
        const float t2 = trip_d(D[(size_t)j * B + j] - dab, soft, alpha, margin);      // anchor j in embed2
        G[(size_t)a * B + j] = -(t1 + t2) * sc;
        diag += t1 + trip_d(daa - D[(size_t)j * B + a], soft, alpha, margin);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) diag += __shfl_xor(diag, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = diag;
    __syncthreads();
    if (threadIdx.x == 0) G[(size_t)a * B + a] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) * sc;
}

// D[a][b] = |e1_a - e2_b|^2: de1[a] = 2 * sum_b G[a][b] (e1_a - e2_b)   (side 0)
//                            de2[b] = 2 * sum_a G[a][b] (e2_b - e1_a)   (side 1); one block per output row.
__global__ __launch_bounds__(256) void sqdist_bwd_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                          const float* __restrict__ G, float* __restrict__ out, int B, int n,
                                                          int side) {
    const int r = blockIdx.x;
    const float* self = (side == 0 ? e1 : e2) + (size_t)r * n;
    const float* other = side == 0 ? e2 : e1;
    for (int k = threadIdx.x; k < n; k += 256) {
        float s = 0.f;
        const float sv = self[k];
        for (int q = 0; q < B; ++q) {
            const float gq = side == 0 ? G[(size_t)r * B + q] : G[(size_t)q * B + r];
            s += gq * (sv - other[(size_t)q * n + k]);
        }
        out[(size_t)r * n + k] = 2.f * s;
    }
}


// Conv2d(k=4, s=2) filter <-> the 3x3 filter over space-to-depth(2) channels the MFMA conv kernels take
// (model/cvig_baseline.py:236-252: channel (dy*2+dx)*ci + c of the space-to-depth image is pixel (2y+dy, 2x+dx), so tap (a+1, b+1),
// a, b in {0,1}, holds w[co][c][2a+dy][2b+dx]; tap row 0 and tap column 0 are zero). One launch each way instead of a zero fill and
// four strided copies per layer and step.
__global__ void conv4x4_to_k3_kernel(const float* __restrict__ w, float* __restrict__ k3, int ci, int cpad, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int tap = idx % 9;
    size_t t = idx / 9;
    const int ch = t % cpad;
    const size_t co = t / cpad;
    const int kh = tap / 3, kw = tap % 3;
    float v = 0.f;
    if (kh >= 1 && kw >= 1 && ch < 4 * ci) {
        const int d = ch / ci, c = ch - d * ci;
        v = w[((co * ci + c) * 4 + 2 * (kh - 1) + (d >> 1)) * 4 + 2 * (kw - 1) + (d & 1)];
    }
    k3[idx] = v;
}

__global__ void k3_to_conv4x4_kernel(const float* __restrict__ k3, float* __restrict__ w, int ci, int cpad, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kx = idx & 3, ky = (idx >> 2) & 3;
    const size_t t = idx >> 4;
    const int c = t % ci;
    const size_t co = t / ci;
    const int ch = ((ky & 1) * 2 + (kx & 1)) * ci + c;
    w[idx] = k3[((co * cpad + ch) * 3 + (ky >> 1) + 1) * 3 + (kx >> 1) + 1];
}

}  // namespace

extern "C" {

int witw_space_to_depth2(const float* x, float* y, int B, int Hp, int Wp, int H, int W, int C, int Cpad, int in_nchw,
                         int normalize, const float* scale, const float* shift, void* stream) {
    WITW_CHECK_ARG(x && y, "space_to_depth2: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp && Cpad >= 4 * C, "space_to_depth2: bad shape");
    const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * Cpad;
    WITW_CHECK_ARG((total + 255) / 256 <= 0x7fffffffULL, "space_to_depth2: tensor too large");
    if (!in_nchw && !normalize && (C & 3) == 0 && Cpad == 4 * C)
        hipLaunchKernelGGL(space_to_depth2_quad_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                           y, Hp, Wp, H, W, C, total / 4, scale, shift);
    else
        hipLaunchKernelGGL(space_to_depth2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, B,
                           Hp, Wp, H, W, C, Cpad, in_nchw, normalize, total, scale, shift);
    WITW_CHECK_LAUNCH("space_to_depth2");
    return WITW_OK;
}

int witw_space_to_depth2_mosaic(const float* x, float* y, int B, int H, int W, int C, int g, void* stream) {
    WITW_CHECK_ARG(x && y, "space_to_depth2_mosaic: null pointer");
    WITW_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && g >= 1 && g <= 64, "space_to_depth2_mosaic: bad shape");
    const size_t total4 = (size_t)cdiv(B, g * g) * (g * ((H + 1) / 2)) * (g * ((W + 1) / 2)) * C;     // 4C / 4
    WITW_CHECK_ARG((total4 + 255) / 256 <= 0x7fffffffULL, "space_to_depth2_mosaic: tensor too large");
    hipLaunchKernelGGL(s2d_mosaic_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W,
                       C, g, total4);
    WITW_CHECK_LAUNCH("space_to_depth2_mosaic");
    return WITW_OK;
}

// ws: the [ksplit][Bm, g*h, g*w, C] partial sums witw_conv3x3_fwd_taps4_ex(ksplit > 1) wrote for B images laid out as g x g
// mosaics of h x w maps (g = 1: plain batch); y [B, vh, vw, C] = the (vh, vw) valid outputs of every image after bias, activation
// (0 none, 1 ReLU, 2 LeakyReLU) and the optional per-channel affine.
int witw_taps4_splitk_finish(const float* ws, int ksplit, const float* bias, int act, float lrelu_slope, const float* post_scale,
                             const float* post_shift, float* y, int B, int g, int h, int w, int vh, int vw, int C, void* stream) {
    WITW_CHECK_ARG(ws && bias && y, "taps4_splitk_finish: null pointer");
    WITW_CHECK_ARG(B > 0 && g >= 1 && h > 0 && w > 0 && vh > 0 && vw > 0 && vh <= h && vw <= w && C > 0 && (C & 3) == 0 && ksplit >= 1,
                   "taps4_splitk_finish: bad shape B=%d g=%d map %dx%d valid %dx%d C=%d ksplit=%d", B, g, h, w, vh, vw, C, ksplit);
    WITW_CHECK_ARG(act >= 0 && act <= 2, "taps4_splitk_finish: activation %d unknown", act);
    WITW_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "taps4_splitk_finish: post_scale and post_shift go together");
    const size_t stride = (size_t)cdiv(B, g * g) * (g * h) * (g * w) * C;
    const size_t total4 = (size_t)B * vh * vw * (C / 4);
    WITW_CHECK_ARG((total4 + 255) / 256 <= 0x7fffffffULL, "taps4_splitk_finish: tensor too large");
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, ksplit,
                       stride, bias, act, lrelu_slope, post_scale, post_shift, y, B, g, h, w, vh, vw, C, total4);
    WITW_CHECK_LAUNCH("taps4_splitk_finish");
    return WITW_OK;
}

int witw_gem_pool(const float* x, float* f, int B, int Hp, int Wp, int H, int W, int C, int ldf, int col0, float p,
                  const float* scale, const float* shift, void* stream) {
    WITW_CHECK_ARG(x && f, "gem_pool: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp && col0 >= 0 && col0 + C <= ldf, "gem_pool: bad shape");
    hipLaunchKernelGGL(gem_pool_kernel, dim3(B, cdiv(C, 64)), dim3(64 * GEM_PH), 0, (hipStream_t)stream, x, f, B, Hp, Wp, H, W, C, ldf,
                       col0, p, scale, shift);
    WITW_CHECK_LAUNCH("gem_pool");
    return WITW_OK;
}

int witw_embed_normalize(float* f, int B, int n, void* stream) {
    WITW_CHECK_ARG(f && B > 0 && n > 0, "embed_normalize: bad argument");
    hipLaunchKernelGGL(embed_normalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f, n);
    WITW_CHECK_LAUNCH("embed_normalize");
    return WITW_OK;
}

int witw_pairwise_sqdist(const float* a, const float* b, float* D, int Na, int Nb, int n, int take_sqrt, void* stream) {
    WITW_CHECK_ARG(a && b && D, "pairwise_sqdist: null pointer");
    WITW_CHECK_ARG(Na > 0 && Nb > 0 && n > 0 && n <= 12288 && Na <= 65535, "pairwise_sqdist: bad shape Na=%d Nb=%d n=%d", Na, Nb, n);
    hipLaunchKernelGGL(pairwise_sqdist_kernel, dim3(cdiv(Nb, 256), Na), dim3(256), n * sizeof(float), (hipStream_t)stream, a, b, D,
                       Na, Nb, n, take_sqrt);
    WITW_CHECK_LAUNCH("pairwise_sqdist");
    return WITW_OK;
}

// exhaustive_minibatch_triplet_loss from D[i][j] = |embed1_i - embed2_j|^2; workspace: B floats.
int witw_exhaustive_triplet_loss(const float* D, int B, int soft_margin, float alpha, float margin, float* loss, float* workspace,
                                 void* stream) {
    WITW_CHECK_ARG(D && loss && workspace, "exhaustive_triplet_loss: null pointer");
    WITW_CHECK_ARG(B >= 2, "exhaustive_triplet_loss: batch %d < 2", B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(exhaustive_partials_kernel, dim3(B), dim3(256), 0, st, D, workspace, B, soft_margin, alpha, margin);
    hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, workspace, loss, B, 2.f * B * (B - 1));
    WITW_CHECK_LAUNCH("exhaustive_triplet_loss");
    return WITW_OK;
}

// ---- training-mode entry points of cvig_baseline
static int bl_rows(size_t npix) {
    size_t r = (npix + 127) / 128;
    return (int)(r < 64 ? 64 : r);
}

// row-based sums (channel_sums_rows_kernel): image rows per block for ~1024 blocks, and whether C qualifies
static int bl_rows_per_block(int nrows) { return cdiv(nrows, 1024); }
static bool bl_wide(int C) { return (C % 4) == 0 && C >= 4 && C <= 1024 && (256 % (C / 4)) == 0; }

// launches the partial sums; returns the number of partials written
static int bl_launch_sums(const float* a, const float* g, const float* mean, const float* invstd, float* part, int B, int Hp,
                          int Wp, int H, int W, int C, hipStream_t st, int g_cp = 0) {
    const size_t npix = (size_t)B * H * W;
    if (bl_wide(C)) {
        const int nrows = B * H, rpb = bl_rows_per_block(nrows), nparts = cdiv(nrows, rpb);
        hipLaunchKernelGGL(channel_sums_rows_kernel, dim3(nparts), dim3(256), 0, st, a, g, mean, invstd, part, Hp, Wp, H, W, C,
                           nrows, rpb, g_cp);
        return nparts;
    }
    const int rows = bl_rows(npix), nparts = (int)((npix + rows - 1) / rows);
    hipLaunchKernelGGL(channel_sums_kernel, dim3(cdiv(C, 64), nparts), dim3(256), 0, st, a, g, mean, invstd, part, Hp, Wp, H, W, C,
                       npix, rows, g_cp);
    return nparts;
}

long long witw_bn_workspace_floats(int B, int H, int W, int C) {
    const size_t npix = (size_t)B * H * W;
    const int rows = bl_rows(npix);
    long long nparts = (long long)((npix + rows - 1) / rows);
    if (nparts < 1024) nparts = 1024;          // the row-based form writes at most 1024 partials
    return nparts * 2 * C + 2 * C;
}

// Batch statistics of BatchNorm2d over the valid region of a [B,Hp,Wp,C] tensor -> mean, invstd, and the affine
// (scale, shift) with y = a*scale + shift; running stats updated in place when non-NULL (momentum, unbiased var).
int witw_bn_train_stats(const float* a, int B, int Hp, int Wp, int H, int W, int C, const float* gamma, const float* beta,
                        float eps, float momentum, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                        float* running_var, float* workspace, void* stream) {
    WITW_CHECK_ARG(a && gamma && beta && mean && invstd && scale && shift && workspace, "bn_train_stats: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp, "bn_train_stats: bad shape");
    WITW_CHECK_ARG((size_t)B * H * W > 1, "bn_train_stats: needs more than one value per channel");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    const int nparts = bl_launch_sums(a, nullptr, nullptr, nullptr, workspace, B, Hp, Wp, H, W, C, st);
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(cdiv(C, BNF_CH)), dim3(BNF_CH * BNF_LANES), 0, st, workspace, nparts, C, (float)npix, gamma, beta,
                       eps, momentum, mean, invstd, scale, shift, running_mean, running_var);
    WITW_CHECK_LAUNCH("bn_train_stats");
    return WITW_OK;
}

// Backward of y = BN_train(lrelu(z)) w.r.t. z, gamma, beta: a = lrelu(z) (saved), dy = gradient at y.
// dy_s2d_cp > 0: dy is still the space-to-depth image the next block's data-gradient conv wrote ([B, ceil(H/2), ceil(W/2), dy_s2d_cp],
// channel ((h&1)*2+(w&1))*C + c) and is read in place (no witw_depth_to_space2 pass); 0: dy is [B,Hp,Wp,C] like a.
int witw_bn_lrelu_bwd_ex(const float* a, const float* dy, float* dz, float* dgamma, float* dbeta, const float* mean,
                         const float* invstd, const float* gamma, int B, int Hp, int Wp, int H, int W, int C, float slope,
                         int dy_s2d_cp, float* workspace, void* stream) {
    WITW_CHECK_ARG(a && dy && dz && dgamma && dbeta && mean && invstd && gamma && workspace, "bn_lrelu_bwd: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp, "bn_lrelu_bwd: bad shape");
    WITW_CHECK_ARG(dy_s2d_cp == 0 || (dy_s2d_cp >= 4 * C && (dy_s2d_cp % 4) == 0), "bn_lrelu_bwd: space-to-depth channel stride %d for C=%d", dy_s2d_cp, C);
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    const int nparts = bl_launch_sums(a, dy, mean, invstd, workspace, B, Hp, Wp, H, W, C, st, dy_s2d_cp);
    float* sums = workspace + (size_t)nparts * 2 * C;
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(cdiv(C, BNF_CH)), dim3(BNF_CH * BNF_LANES), 0, st, workspace, nparts, C, sums, dgamma, dbeta);
    const size_t total = (size_t)B * Hp * Wp * C;
    if ((C & 3) == 0)
        hipLaunchKernelGGL(bn_lrelu_bwd_apply_quad_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, a, dy, dz, mean,
                           invstd, gamma, sums, Hp, Wp, H, W, C, (float)npix, slope, total / 4, dy_s2d_cp);
    else
        hipLaunchKernelGGL(bn_lrelu_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, dy, dz, mean, invstd,
                           gamma, sums, Hp, Wp, H, W, C, (float)npix, slope, total, dy_s2d_cp);
    WITW_CHECK_LAUNCH("bn_lrelu_bwd");
    return WITW_OK;
}

int witw_bn_lrelu_bwd(const float* a, const float* dy, float* dz, float* dgamma, float* dbeta, const float* mean,
                      const float* invstd, const float* gamma, int B, int Hp, int Wp, int H, int W, int C, float slope,
                      float* workspace, void* stream) {
    return witw_bn_lrelu_bwd_ex(a, dy, dz, dgamma, dbeta, mean, invstd, gamma, B, Hp, Wp, H, W, C, slope, 0, workspace, stream);
}

int witw_depth_to_space2(const float* g, const float* add, float* dx, int B, int Hp, int Wp, int H, int W, int C, int Cpad,
                         void* stream) {
    WITW_CHECK_ARG(g && dx, "depth_to_space2: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp && Cpad >= 4 * C, "depth_to_space2: bad shape");
    const size_t total = (size_t)B * Hp * Wp * C;
    hipLaunchKernelGGL(depth_to_space2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, add, dx,
                       Hp, Wp, H, W, C, Cpad, total);
    WITW_CHECK_LAUNCH("depth_to_space2");
    return WITW_OK;
}

int witw_gem_pool_bwd(const float* a, const float* scale, const float* shift, const float* f, const float* df, float* dy, int B,
                      int Hp, int Wp, int H, int W, int C, int ldf, int col0, float p, int accumulate, void* stream) {
    WITW_CHECK_ARG(a && scale && shift && f && df && dy, "gem_pool_bwd: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp, "gem_pool_bwd: bad shape");
    const size_t total = (size_t)B * Hp * Wp * C;
    hipLaunchKernelGGL(gem_pool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, scale, shift,
                       f, df, dy, Hp, Wp, H, W, C, ldf, col0, p, accumulate, total);
    WITW_CHECK_LAUNCH("gem_pool_bwd");
    return WITW_OK;
}

// g: the un-normalised concat feature [B,n]; df: gradient at f = g/|g|^(1/2); dg out.
int witw_embed_normalize_bwd(const float* g, const float* df, float* dg, int B, int n, void* stream) {
    WITW_CHECK_ARG(g && df && dg && B > 0 && n > 0, "embed_normalize_bwd: bad argument");
    hipLaunchKernelGGL(embed_normalize_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, g, df, dg, n);
    WITW_CHECK_LAUNCH("embed_normalize_bwd");
    return WITW_OK;
}

// backward of witw_pairwise_sqdist + witw_exhaustive_triplet_loss: grad_loss [1] -> de1, de2 [B,n]; workspace B*B floats.
int witw_exhaustive_triplet_loss_bwd(const float* e1, const float* e2, const float* D, const float* grad_loss, float* de1,
                                     float* de2, int B, int n, int soft_margin, float alpha, float margin, float* workspace,
                                     void* stream) {
    WITW_CHECK_ARG(e1 && e2 && D && grad_loss && de1 && de2 && workspace, "exhaustive_triplet_loss_bwd: null pointer");
    WITW_CHECK_ARG(B >= 2 && n > 0, "exhaustive_triplet_loss_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(exhaustive_bwd_kernel, dim3(B), dim3(256), 0, st, D, grad_loss, workspace, B, soft_margin, alpha, margin);
    hipLaunchKernelGGL(sqdist_bwd_kernel, dim3(B), dim3(256), 0, st, e1, e2, workspace, de1, B, n, 0);
    hipLaunchKernelGGL(sqdist_bwd_kernel, dim3(B), dim3(256), 0, st, e1, e2, workspace, de2, B, n, 1);
    WITW_CHECK_LAUNCH("exhaustive_triplet_loss_bwd");
    return WITW_OK;
}

// w [co][ci][4][4] (torch layout of Conv2d(ci, co, 4, 2)) -> k3 [co][cpad][3][3], cpad >= 4*ci (extra channels zero)
int witw_conv4x4_to_k3(const float* w, float* k3, int co, int ci, int cpad, void* stream) {
    WITW_CHECK_ARG(w && k3 && co > 0 && ci > 0 && cpad >= 4 * ci, "conv4x4_to_k3: bad arguments");
    const size_t total = (size_t)co * cpad * 9;
    hipLaunchKernelGGL(conv4x4_to_k3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, k3, ci, cpad, total);
    WITW_CHECK_LAUNCH("conv4x4_to_k3");
    return WITW_OK;
}

// the inverse gather (a weight gradient in the k3 layout -> the Conv2d parameter's layout)
int witw_k3_to_conv4x4(const float* k3, float* w, int co, int ci, int cpad, void* stream) {
    WITW_CHECK_ARG(w && k3 && co > 0 && ci > 0 && cpad >= 4 * ci, "k3_to_conv4x4: bad arguments");
    const size_t total = (size_t)co * ci * 16;
    hipLaunchKernelGGL(k3_to_conv4x4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k3, w, ci, cpad, total);
    WITW_CHECK_LAUNCH("k3_to_conv4x4");
    return WITW_OK;
}

}  // extern "C"
