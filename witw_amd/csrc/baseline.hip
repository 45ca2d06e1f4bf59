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
                                       int C, int Cp, int in_nchw, int normalize, size_t total) {
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
        }
    }
    y[idx] = v;
}

// f[b, col0 + c] = (mean_{h<H,w<W} relu(x[b,h,w,c])^p)^(1/p); x NHWC [B,Hp,Wp,C]; one thread per (b,c).
__global__ void gem_pool_kernel(const float* __restrict__ x, float* __restrict__ f, int B, int Hp, int Wp, int H, int W, int C,
                                int ldf, int col0, float p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int c = idx % C, b = idx / C;
    float s = 0.f;
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            const float v = fmaxf(x[(((size_t)b * Hp + h) * Wp + w) * C + c], 0.f);
            s += powf(v, p);
        }
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

}  // namespace

extern "C" {

int witw_space_to_depth2(const float* x, float* y, int B, int Hp, int Wp, int H, int W, int C, int Cpad, int in_nchw,
                         int normalize, void* stream) {
    WITW_CHECK_ARG(x && y, "space_to_depth2: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp && Cpad >= 4 * C, "space_to_depth2: bad shape");
    const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * Cpad;
    WITW_CHECK_ARG((total + 255) / 256 <= 0x7fffffffULL, "space_to_depth2: tensor too large");
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, B,
                       Hp, Wp, H, W, C, Cpad, in_nchw, normalize, total);
    WITW_CHECK_LAUNCH("space_to_depth2");
    return WITW_OK;
}

int witw_gem_pool(const float* x, float* f, int B, int Hp, int Wp, int H, int W, int C, int ldf, int col0, float p,
                  void* stream) {
    WITW_CHECK_ARG(x && f, "gem_pool: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp && col0 >= 0 && col0 + C <= ldf, "gem_pool: bad shape");
    hipLaunchKernelGGL(gem_pool_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, x, f, B, Hp, Wp, H, W, C, ldf,
                       col0, p);
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

}  // extern "C"
