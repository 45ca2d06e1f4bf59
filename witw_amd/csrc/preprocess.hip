// Device data path of the reference's host transforms (gfx950, HBM-bound byte/pixel work):
//   Resize              model/cvig_fov.py:117-134 (torchvision F.resize on float CHW = bilinear,
//                       align_corners=False, no antialias under the pinned torchvision 0.9.1)
//   ImageNormalization  model/cvig_fov.py:137-149  ((x/255 - mean)/std; semantic variant
//                       model/cvig_semantic.py:172-176 divides only channels 0-2 by 255)
//   PolarTransform      model/cvig_fov.py:186-209 + bilinear_interpolate :156-183
// Compiled with -ffp-contract=off so products and sums round exactly like the reference's
// separate elementwise torch ops (polar: wa*Ia + wb*Ib + wc*Ic + wd*Id, left to right).
#include "common.h"

namespace {

struct NormArgs {
    float mean[8];
    float stdv[8];
    int n_div255;   // channels [0,n_div255) are divided by 255 first
    int enabled;
};

// One thread per output pixel (b, y, x); loops over channels. Source index / lambda follow
// ATen's area_pixel_compute_source_index (align_corners=False): src = scale*(dst+0.5)-0.5, <0 -> 0.
__global__ void resize_bilinear_norm_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int Hi, int Wi,
                                            int Ho, int Wo, float sh, float sw, NormArgs na) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo;
    if (idx >= total) return;
    const int ox = idx % Wo;
    size_t t = idx / Wo;
    const int oy = t % Ho;
    const int b = (int)(t / Ho);
    // ATen's CPU build contracts scale*(dst+0.5)-0.5 into one fma; follow it (the last bit of the
    // source coordinate moves lambda by ~1e-5, i.e. up to 4e-3 on 0..255 pixel values)
    float fy = __fmaf_rn(sh, oy + 0.5f, -0.5f);
    if (fy < 0.f) fy = 0.f;
    float fx = __fmaf_rn(sw, ox + 0.5f, -0.5f);
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < Hi - 1) ? 1 : 0), x1 = x0 + ((x0 < Wi - 1) ? 1 : 0);
    const float ly1 = fy - y0, lx1 = fx - x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    for (int c = 0; c < C; ++c) {
        const float* p = x + ((size_t)b * C + c) * Hi * Wi;
        const float top = lx0 * p[(size_t)y0 * Wi + x0] + lx1 * p[(size_t)y0 * Wi + x1];
        const float bot = lx0 * p[(size_t)y1 * Wi + x0] + lx1 * p[(size_t)y1 * Wi + x1];
        float v = ly0 * top + ly1 * bot;
        if (na.enabled) {
            if (c < na.n_div255) v = v / 255.f;
            v = (v - na.mean[c]) / na.stdv[c];
        }
        y[(((size_t)b * C + c) * Ho + oy) * Wo + ox] = v;
    }
}

__global__ void normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int C, size_t hw, size_t total,
                                 NormArgs na) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (idx / hw) % C;
    float v = x[idx];
    if (c < na.n_div255) v = v / 255.f;
    y[idx] = (v - na.mean[c]) / na.stdv[c];
}

// taps: int4 {ia, ib, ic, id} flat offsets into one size x size plane; wts: float4 {wa, wb, wc, wd}.
// One thread per (b, output pixel); the 32-byte LUT entry is read once and reused for all channels.
__global__ void polar_kernel(const float* __restrict__ x, const int4* __restrict__ taps, const float4* __restrict__ wts,
                             float* __restrict__ y, int B, int C, int plane_in, int plane_out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * plane_out) return;
    const int pix = idx % plane_out;
    const int b = (int)(idx / plane_out);
    const int4 t = taps[pix];
    const float4 w = wts[pix];
    for (int c = 0; c < C; ++c) {
        const float* p = x + ((size_t)b * C + c) * plane_in;
        const float v = ((w.x * p[t.x] + w.y * p[t.y]) + w.z * p[t.z]) + w.w * p[t.w];
        y[((size_t)b * C + c) * plane_out + pix] = v;
    }
}

int fill_norm(NormArgs& na, int C, const float* mean, const float* stdv, int n_div255) {
    na.enabled = (mean != nullptr && stdv != nullptr);
    na.n_div255 = n_div255;
    for (int c = 0; c < 8; ++c) {
        na.mean[c] = (na.enabled && c < C) ? mean[c] : 0.f;
        na.stdv[c] = (na.enabled && c < C) ? stdv[c] : 1.f;
    }
    return 0;
}

}  // namespace

extern "C" {

// mean/stdv are HOST arrays of C floats (nullptr: resize only).
int witw_resize_bilinear_normalize(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, const float* mean,
                                   const float* stdv, int n_div255, void* stream) {
    WITW_CHECK_ARG(x && y, "resize: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0,
                   "resize: bad shape B=%d C=%d %dx%d -> %dx%d", B, C, Hi, Wi, Ho, Wo);
    NormArgs na;
    fill_norm(na, C, mean, stdv, n_div255);
    const size_t total = (size_t)B * Ho * Wo;
    hipLaunchKernelGGL(resize_bilinear_norm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       y, B, C, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo, na);
    WITW_CHECK_LAUNCH("resize_bilinear_normalize");
    return WITW_OK;
}

int witw_normalize(const float* x, float* y, int B, int C, int H, int W, const float* mean, const float* stdv, int n_div255,
                   void* stream) {
    WITW_CHECK_ARG(x && y && mean && stdv, "normalize: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "normalize: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    NormArgs na;
    fill_norm(na, C, mean, stdv, n_div255);
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, C,
                       (size_t)H * W, total, na);
    WITW_CHECK_LAUNCH("normalize");
    return WITW_OK;
}

// taps/wts: DEVICE LUTs of Ho*Wo entries built on the host in fp64 exactly as the reference does.
int witw_polar_transform(const float* x, const int* taps, const float* wts, float* y, int B, int C, int size, int Ho, int Wo,
                         void* stream) {
    WITW_CHECK_ARG(x && taps && wts && y, "polar_transform: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && size > 0 && Ho > 0 && Wo > 0, "polar_transform: bad shape");
    const size_t total = (size_t)B * Ho * Wo;
    hipLaunchKernelGGL(polar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (const int4*)taps, (const float4*)wts, y, B, C, size * size, Ho * Wo);
    WITW_CHECK_LAUNCH("polar_transform");
    return WITW_OK;
}

}  // extern "C"
