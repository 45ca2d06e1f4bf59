// Device data path of the reference's host transforms (gfx950, HBM-bound byte/pixel work):
//   Resize              model/cvig_fov.py:117-134 (torchvision F.resize on float CHW = bilinear,
//                       align_corners=False, no antialias under the pinned torchvision 0.9.1)
//   ImageNormalization  model/cvig_fov.py:137-149  ((x/255 - mean)/std; semantic variant
//                       model/cvig_semantic.py:172-176 divides only channels 0-2 by 255)
//   PolarTransform      model/cvig_fov.py:186-209 + bilinear_interpolate :156-183
//   SyncedRotation      model/cvig_baseline.py:131-144 (torchvision F.rotate on a float CHW tensor = affine grid +
//                       grid_sample(nearest, zeros, align_corners=False) under torchvision 0.9.1)
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

// The same resize (+ normalisation) for a BATCH of images of different sizes in one launch: image b is described by
// desc[b] = {address, H, W, start, pixel layout}; KIND 0 = float32 planar CHW (what ImagePairDataset returns, reference
// model/cvig_fov.py:88-95), KIND 1 = uint8 interleaved HWC straight from the decoder (a quarter of the bytes over PCIe),
// converted to float exactly (0..255). The panorama branch of Resize (:118-128: resize to Wfull columns, then keep Wo of them
// from a per-sample start, wrapping) is the column map ox -> (ox + start) mod Wfull. Same arithmetic, operation for
// operation, as resize_bilinear_norm_kernel.
struct ImgDesc {
    unsigned long long ptr;
    long long H, W, start, cs;      // cs: channels stored per pixel (KIND 1) -- the first C are used
};

template <int KIND>
__global__ void resize_batched_kernel(const ImgDesc* __restrict__ desc, float* __restrict__ y, int B, int C, int Ho, int Wo,
                                      int Wfull, NormArgs na) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo;
    if (idx >= total) return;
    const int ox = idx % Wo;
    size_t t = idx / Wo;
    const int oy = t % Ho;
    const int b = (int)(t / Ho);
    const ImgDesc d = desc[b];
    const int Hi = (int)d.H, Wi = (int)d.W;
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wfull;
    int cx = ox + (int)d.start;
    if (cx >= Wfull) cx -= Wfull;
    float fy = __fmaf_rn(sh, oy + 0.5f, -0.5f);
    if (fy < 0.f) fy = 0.f;
    float fx = __fmaf_rn(sw, cx + 0.5f, -0.5f);
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < Hi - 1) ? 1 : 0), x1 = x0 + ((x0 < Wi - 1) ? 1 : 0);
    const float ly1 = fy - y0, lx1 = fx - x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    for (int c = 0; c < C; ++c) {
        float p00, p01, p10, p11;
        if (KIND == 0) {
            const float* p = reinterpret_cast<const float*>(d.ptr) + (size_t)c * Hi * Wi;
            p00 = p[(size_t)y0 * Wi + x0]; p01 = p[(size_t)y0 * Wi + x1];
            p10 = p[(size_t)y1 * Wi + x0]; p11 = p[(size_t)y1 * Wi + x1];
        } else {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(d.ptr) + c;
            const size_t cs = (size_t)d.cs;
            p00 = (float)p[((size_t)y0 * Wi + x0) * cs]; p01 = (float)p[((size_t)y0 * Wi + x1) * cs];
            p10 = (float)p[((size_t)y1 * Wi + x0) * cs]; p11 = (float)p[((size_t)y1 * Wi + x1) * cs];
        }
        const float top = lx0 * p00 + lx1 * p01;
        const float bot = lx0 * p10 + lx1 * p11;
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

// theta: per image the 2x3 inverse rotation matrix ALREADY divided by (0.5*W, 0.5*H) as torchvision's
// _gen_affine_grid does, laid out [b][k][j] (k = x,y,1 row; j = output coordinate): gx = x*t00 + y*t10 + t20.
// Base grid = half-integer pixel centres (linspace(-W/2+.5, W/2-.5, W): step exactly 1); un-normalisation
// and rounding follow ATen's grid_sampler (align_corners=False, nearbyint, zero outside).
__global__ void rotate_nearest_kernel(const float* __restrict__ x, const float* __restrict__ theta, float* __restrict__ y,
                                      int B, int C, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * H * W) return;
    const int ox = idx % W;
    size_t t = idx / W;
    const int oy = t % H;
    const int b = (int)(t / H);
    const float* th = theta + (size_t)b * 6;
    const float xg = (float)ox + (0.5f - 0.5f * (float)W), yg = (float)oy + (0.5f - 0.5f * (float)H);
    const float gx = __fmaf_rn(yg, th[2], xg * th[0]) + th[4];
    const float gy = __fmaf_rn(yg, th[3], xg * th[1]) + th[5];
    const float fx = ((gx + 1.f) * (float)W - 1.f) / 2.f, fy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float rx = nearbyintf(fx), ry = nearbyintf(fy);
    const bool ok = rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1);
    const size_t src = ok ? (size_t)(int)ry * W + (int)rx : 0;
    for (int c = 0; c < C; ++c) {
        const size_t plane = ((size_t)b * C + c) * H * W;
        y[plane + (size_t)oy * W + ox] = ok ? x[plane + src] : 0.f;
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

// Resize (+ normalise) B images of individual sizes in ONE launch. desc: DEVICE array [B][5] of 64-bit words {device address of
// the image, H, W, start column, channels stored per pixel}; kind 0 = float32 planar CHW sources, 1 = uint8 interleaved HWC.
// Output [B,C,Ho,Wo]: the image resized to Ho x Wfull, of which the Wo columns from `start` on (wrapping) are kept
// (Wfull == Wo, start == 0: a plain resize). mean / stdv as in witw_resize_bilinear_normalize.
int witw_resize_bilinear_normalize_batched(const void* desc, float* y, int B, int C, int Ho, int Wo, int Wfull, int kind,
                                           const float* mean, const float* stdv, int n_div255, void* stream) {
    WITW_CHECK_ARG(desc && y, "resize_batched: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && C <= 8 && Ho > 0 && Wo > 0 && Wfull >= Wo, "resize_batched: bad shape B=%d C=%d -> %dx%d of %d",
                   B, C, Ho, Wo, Wfull);
    WITW_CHECK_ARG(kind == 0 || kind == 1, "resize_batched: source kind %d unknown (0 float32 CHW, 1 uint8 HWC)", kind);
    NormArgs na;
    fill_norm(na, C, mean, stdv, n_div255);
    const size_t total = (size_t)B * Ho * Wo;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (kind == 0)
        hipLaunchKernelGGL(resize_batched_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const ImgDesc*)desc, y, B, C, Ho, Wo,
                           Wfull, na);
    else
        hipLaunchKernelGGL(resize_batched_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const ImgDesc*)desc, y, B, C, Ho, Wo,
                           Wfull, na);
    WITW_CHECK_LAUNCH("resize_bilinear_normalize_batched");
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

// The same 4-tap gather for any sampling table (bilinear_interpolate of the reference, model/cvig_fov.py:156-183, on
// arbitrary coordinates): taps are flat offsets into a plane of plane_in elements, n_out output samples per plane.
int witw_bilinear_gather(const float* x, const int* taps, const float* wts, float* y, int B, int C, long long plane_in,
                         long long n_out, void* stream) {
    WITW_CHECK_ARG(x && taps && wts && y, "bilinear_gather: null pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && plane_in > 0 && plane_in <= 0x7fffffffLL && n_out > 0 && n_out <= 0x7fffffffLL,
                   "bilinear_gather: bad shape");
    const size_t total = (size_t)B * n_out;
    hipLaunchKernelGGL(polar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (const int4*)taps, (const float4*)wts, y, B, C, (int)plane_in, (int)n_out);
    WITW_CHECK_LAUNCH("bilinear_gather");
    return WITW_OK;
}

// x, y: [B,C,H,W] fp32 (y != x); theta: DEVICE [B,2,3]^T rescaled matrices, see rotate_nearest_kernel.
int witw_rotate_nearest(const float* x, const float* theta, float* y, int B, int C, int H, int W, void* stream) {
    WITW_CHECK_ARG(x && theta && y && x != y, "rotate_nearest: null or aliased pointer");
    WITW_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "rotate_nearest: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(rotate_nearest_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       theta, y, B, C, H, W);
    WITW_CHECK_LAUNCH("rotate_nearest");
    return WITW_OK;
}

}  // extern "C"
