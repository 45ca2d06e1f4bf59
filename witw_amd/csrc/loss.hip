// Soft-margin triplet loss over the B x B distance matrix (gfx950).
//
// Reference: triplet_loss, model/cvig_fov.py:366-382:
//   loss = ( sum_ij log(1+exp(a*(d_jj - d_ij))) + sum_ij log(1+exp(a*(d_ii - d_ij))) ) / (2B(B-1))
// (both sums over the FULL matrix, diagonal included; naive log(1+exp) kept).
// All reductions run in a fixed order (per-row / per-column partials, then one block) so the
// result is bitwise reproducible run to run.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// blocks [0,B): row i  -> ws[i]      = sum_j softplus(a*(d_ii - d_ij)), ws[2B+i] = sum_j sigmoid(.)
// blocks [B,2B): col j -> ws[B+j]    = sum_i softplus(a*(d_jj - d_ij)), ws[3B+j] = sum_i sigmoid(.)
__global__ __launch_bounds__(256) void triplet_partials_kernel(const float* __restrict__ D, float* __restrict__ ws, int B,
                                                                float alpha) {
    __shared__ float sh[4];
    const bool is_col = blockIdx.x >= (unsigned)B;
    const int m = is_col ? blockIdx.x - B : blockIdx.x;
    const float dm = D[(size_t)m * B + m];
    float sp = 0.f, sg = 0.f;
    for (int t = threadIdx.x; t < B; t += 256) {
        const float d = is_col ? D[(size_t)t * B + m] : D[(size_t)m * B + t];
        const float x = alpha * (dm - d);
        sp += logf(1.f + expf(x));
        sg += 1.f / (1.f + expf(-x));
    }
    const float tsp = block_sum_256(sp, sh);
    const float tsg = block_sum_256(sg, sh);
    if (threadIdx.x == 0) {
        ws[(is_col ? B : 0) + m] = tsp;
        ws[(is_col ? 3 * B : 2 * B) + m] = tsg;
    }
}

__global__ __launch_bounds__(256) void triplet_finish_kernel(const float* __restrict__ ws, float* __restrict__ loss, int B,
                                                              float norm) {
    __shared__ float sh[4];
    float a = 0.f, b = 0.f;
    for (int t = threadIdx.x; t < B; t += 256) {
        a += ws[B + t];   // surface -> overhead term (column partials), reference :377
        b += ws[t];       // overhead -> surface term (row partials), reference :378
    }
    const float ta = block_sum_256(a, sh);
    const float tb = block_sum_256(b, sh);
    if (threadIdx.x == 0) loss[0] = (ta + tb) / norm;
}

// dL/dD_ij = g*(a/norm) * ( -sig(a(d_jj-d_ij)) - sig(a(d_ii-d_ij)) + [i==j]*(colsig_j + rowsig_i) )
__global__ void triplet_bwd_kernel(const float* __restrict__ D, const float* __restrict__ ws,
                                   const float* __restrict__ gloss, float* __restrict__ gD, int B, float alpha,
                                   float norm) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * B) return;
    const int i = idx / B, j = idx - (size_t)i * B;
    const float d = D[idx];
    const float x1 = alpha * (D[(size_t)j * B + j] - d);
    const float x2 = alpha * (D[(size_t)i * B + i] - d);
    float g = -(1.f / (1.f + expf(-x1))) - (1.f / (1.f + expf(-x2)));
    if (i == j) g += ws[3 * B + j] + ws[2 * B + i];
    gD[idx] = g * (gloss[0] * alpha / norm);
}

// Column-slab form for the sharded global batch: this rank holds D[:, col0:col0+Bs] (all overheads x its
// own surfaces) and the full diagonal. part[i] = sum_j softplus(a(d_{c,c} - d_ij)) + softplus(a(d_ii - d_ij)), c = col0+j.
__global__ __launch_bounds__(256) void triplet_slab_partials_kernel(const float* __restrict__ D, const float* __restrict__ diag,
                                                                     float* __restrict__ part, int Bs, int col0, float alpha) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const float dii = diag[i];
    float s = 0.f;
    for (int j = threadIdx.x; j < Bs; j += 256) {
        const float d = D[(size_t)i * Bs + j];
        s += logf(1.f + expf(alpha * (diag[col0 + j] - d)));
        s += logf(1.f + expf(alpha * (dii - d)));
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[i] = t;
}

// Sigmoid sums of a column slab (backward of the sharded loss):
// blocks [0,Bo): rowsig[i] = sum_{j in slab} sig(a(d_ii - d_ij));  blocks [Bo,Bo+Bs): colsig[j] = sum_i sig(a(d_cc - d_ij)), c = col0+j
__global__ __launch_bounds__(256) void triplet_slab_sig_kernel(const float* __restrict__ D, const float* __restrict__ diag,
                                                                float* __restrict__ rowsig, float* __restrict__ colsig, int Bo, int Bs,
                                                                int col0, float alpha) {
    __shared__ float sh[4];
    const bool is_col = blockIdx.x >= (unsigned)Bo;
    const int m = is_col ? blockIdx.x - Bo : blockIdx.x;
    const float dm = is_col ? diag[col0 + m] : diag[m];
    const int n = is_col ? Bo : Bs;
    float sg = 0.f;
    for (int t = threadIdx.x; t < n; t += 256) {
        const float d = is_col ? D[(size_t)t * Bs + m] : D[(size_t)m * Bs + t];
        sg += 1.f / (1.f + expf(-alpha * (dm - d)));
    }
    const float tot = block_sum_256(sg, sh);
    if (threadIdx.x == 0) (is_col ? colsig : rowsig)[m] = tot;
}

// dL/dD_ij over the slab: g*(a/norm) * ( -sig(a(d_cc-d_ij)) - sig(a(d_ii-d_ij)) + [i==c]*(colsig_j + rowsig_i) ), c = col0+j,
// rowsig = the row sums over ALL surfaces (summed over the ranks by the caller)
__global__ void triplet_slab_bwd_kernel(const float* __restrict__ D, const float* __restrict__ diag, const float* __restrict__ rowsig,
                                        const float* __restrict__ colsig, const float* __restrict__ gloss, float* __restrict__ gD,
                                        int Bo, int Bs, int col0, float alpha, float norm) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Bo * Bs) return;
    const int i = idx / Bs, j = idx - (size_t)i * Bs;
    const float d = D[idx];
    const float x1 = alpha * (diag[col0 + j] - d);
    const float x2 = alpha * (diag[i] - d);
    float g = -(1.f / (1.f + expf(-x1))) - (1.f / (1.f + expf(-x2)));
    if (i == col0 + j) g += colsig[j] + rowsig[i];
    gD[idx] = g * (gloss[0] * alpha / norm);
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, int n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int t = threadIdx.x; t < n; t += 256) s += part[t];
    const float tot = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = tot;
}

}  // namespace

extern "C" {

int witw_triplet_loss_fwd(const float* distance, int B, float alpha, float* loss, float* workspace, void* stream) {
    WITW_CHECK_ARG(distance && loss && workspace, "triplet_loss_fwd: null pointer");
    WITW_CHECK_ARG(B >= 2, "triplet_loss_fwd: batch %d < 2 (the reference divides by 2B(B-1))", B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(triplet_partials_kernel, dim3(2 * B), dim3(256), 0, st, distance, workspace, B, alpha);
    hipLaunchKernelGGL(triplet_finish_kernel, dim3(1), dim3(256), 0, st, workspace, loss, B, 2.f * B * (B - 1));
    WITW_CHECK_LAUNCH("triplet_loss_fwd");
    return WITW_OK;
}

// Un-normalised partial of the loss over a column slab D[Bo][Bs] = distances of all Bo overheads to the
// surfaces [col0, col0+Bs) of the global batch; diag[Bo] = the global diagonal. The caller sums the partials of
// all ranks and divides by 2*Bo*(Bo-1). workspace: Bo floats.
int witw_triplet_loss_slab_fwd(const float* distance, const float* diag, int Bo, int Bs, int col0, float alpha, float* partial,
                               float* workspace, void* stream) {
    WITW_CHECK_ARG(distance && diag && partial && workspace, "triplet_loss_slab_fwd: null pointer");
    WITW_CHECK_ARG(Bo >= 2 && Bs >= 1 && col0 >= 0 && col0 + Bs <= Bo, "triplet_loss_slab_fwd: bad slab Bo=%d Bs=%d col0=%d", Bo, Bs,
                   col0);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(triplet_slab_partials_kernel, dim3(Bo), dim3(256), 0, st, distance, diag, workspace, Bs, col0, alpha);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, st, workspace, partial, Bo);
    WITW_CHECK_LAUNCH("triplet_loss_slab_fwd");
    return WITW_OK;
}

// Backward of the sharded loss, step 1: rowsig[Bo] (partial: this rank's surfaces only — all-reduce it over the
// ranks) and colsig[Bs] (complete) of the slab, see witw_triplet_loss_slab_fwd for the arguments.
int witw_triplet_loss_slab_sig(const float* distance, const float* diag, int Bo, int Bs, int col0, float alpha, float* rowsig,
                               float* colsig, void* stream) {
    WITW_CHECK_ARG(distance && diag && rowsig && colsig, "triplet_loss_slab_sig: null pointer");
    WITW_CHECK_ARG(Bo >= 2 && Bs >= 1 && col0 >= 0 && col0 + Bs <= Bo, "triplet_loss_slab_sig: bad slab Bo=%d Bs=%d col0=%d", Bo, Bs,
                   col0);
    hipLaunchKernelGGL(triplet_slab_sig_kernel, dim3(Bo + Bs), dim3(256), 0, (hipStream_t)stream, distance, diag, rowsig, colsig, Bo,
                       Bs, col0, alpha);
    WITW_CHECK_LAUNCH("triplet_loss_slab_sig");
    return WITW_OK;
}

// Step 2: grad_distance[Bo,Bs] of the slab for grad_loss (device scalar) of the GLOBAL loss; rowsig = the summed row sums.
int witw_triplet_loss_slab_bwd(const float* distance, const float* diag, const float* rowsig, const float* colsig,
                               const float* grad_loss, float* grad_distance, int Bo, int Bs, int col0, float alpha, void* stream) {
    WITW_CHECK_ARG(distance && diag && rowsig && colsig && grad_loss && grad_distance, "triplet_loss_slab_bwd: null pointer");
    WITW_CHECK_ARG(Bo >= 2 && Bs >= 1 && col0 >= 0 && col0 + Bs <= Bo, "triplet_loss_slab_bwd: bad slab Bo=%d Bs=%d col0=%d", Bo, Bs,
                   col0);
    const size_t total = (size_t)Bo * Bs;
    hipLaunchKernelGGL(triplet_slab_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, distance, diag,
                       rowsig, colsig, grad_loss, grad_distance, Bo, Bs, col0, alpha, 2.f * Bo * (Bo - 1));
    WITW_CHECK_LAUNCH("triplet_loss_slab_bwd");
    return WITW_OK;
}

int witw_triplet_loss_bwd(const float* distance, const float* workspace, const float* grad_loss, float* grad_distance, int B,
                          float alpha, void* stream) {
    WITW_CHECK_ARG(distance && workspace && grad_loss && grad_distance, "triplet_loss_bwd: null pointer");
    WITW_CHECK_ARG(B >= 2, "triplet_loss_bwd: batch %d < 2", B);
    const size_t total = (size_t)B * B;
    hipLaunchKernelGGL(triplet_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, distance,
                       workspace, grad_loss, grad_distance, B, alpha, 2.f * B * (B - 1));
    WITW_CHECK_LAUNCH("triplet_loss_bwd");
    return WITW_OK;
}

}  // extern "C"

// ---- Dropout2d masks (model/cvig_fov.py:234-245, p = 0.2 behind convs 17 / 19 / 21): whole channels dropped per sample, the
// kept ones scaled by 1/(1-p). The reference draws them from torch's global RNG stream; here a counter-based generator
// (Philox4x32-10, Salmon et al. SC'11) keyed on the run's seed with the counter (sample*C + channel, layer | encoder << 16,
// step, rank): a mask depends on nothing but those numbers, so an N-rank run is reproducible from (seed, rank, step) whatever
// the launch order, and the backward can re-derive it.
namespace {

__device__ __forceinline__ unsigned philox4x32_10_first(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

__global__ void dropout2d_scales_kernel(float* __restrict__ out, unsigned k0, unsigned k1, unsigned encoder, unsigned step,
                                        unsigned rank, int l0, int l1, int l2, int n_layers, int BC, float p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_layers * BC) return;
    const int li = i / BC, e = i - li * BC;
    const unsigned layer = (unsigned)(li == 0 ? l0 : li == 1 ? l1 : l2);
    const unsigned x = philox4x32_10_first(k0, k1, (unsigned)e, layer | (encoder << 16), step, rank);
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);       // 24 bits -> [0,1)
    out[i] = u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

}  // namespace

extern "C" {

// out [n_layers][B][C]: the Dropout2d scales of up to three layers of one encoder call in one launch.
int witw_dropout2d_scales(float* out, unsigned long long seed, unsigned encoder, unsigned step, unsigned rank, const int* layers,
                          int n_layers, int B, int C, float p, void* stream) {
    WITW_CHECK_ARG(out && layers, "dropout2d_scales: null pointer");
    WITW_CHECK_ARG(n_layers >= 1 && n_layers <= 3 && B > 0 && C > 0, "dropout2d_scales: bad shape layers=%d B=%d C=%d", n_layers, B, C);
    WITW_CHECK_ARG(p >= 0.f && p < 1.f, "dropout2d_scales: p=%f outside [0,1)", (double)p);
    WITW_CHECK_ARG(encoder < 65536u, "dropout2d_scales: encoder id %u too large", encoder);
    const int total = n_layers * B * C;
    hipLaunchKernelGGL(dropout2d_scales_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, (unsigned)seed,
                       (unsigned)(seed >> 32), encoder, step, rank, layers[0], n_layers > 1 ? layers[1] : 0,
                       n_layers > 2 ? layers[2] : 0, n_layers, B * C, p);
    WITW_CHECK_LAUNCH("dropout2d_scales");
    return WITW_OK;
}

}  // extern "C"
