// Back end of the JPEG decoder on the GPU (gfx950; integer / byte work, HBM-bound): what libjpeg does after entropy decoding for
// the images the reference reads in its DataLoader workers (skimage.io.imread -> PIL -> libjpeg-turbo, model/cvig_fov.py:88-89,
// :402). The entropy decoding stays on host threads (witw_amd/csrc_host/jpeg_coef.cpp); the quantised coefficient blocks of a
// whole batch cross PCIe and
//   jpeg_idct_kernel    dequantises and runs libjpeg's 'islow' integer inverse DCT (jidctint.c: 13-bit constants, column pass
//                       then row pass, range limit) -> one 8-bit plane per component,
//   jpeg_to_rgb_kernel  'fancy' chroma upsampling (jdsample.c: h2v1 / h2v2 triangle filter with libjpeg's alternating rounding,
//                       edge rows / columns replicated) + YCbCr -> RGB (jdcolor.c, 16-bit fixed point) -> interleaved u8 HWC,
// the byte layout PIL hands the reference (and witw_resize_bilinear_normalize_batched kind 1 / witw_polar_from_raw read).
// Byte-identical to Pillow's decode: tests/test_jpeg*.py (fixtures under tests/golden/jpeg/ + files written at test time).
#include "common.h"

namespace {

struct JpegPlane {       // one component of one image: int64 x 6 on the host side
    long long coef_blk;  // first block of the component in the batch's coefficient array
    long long qt_idx;    // quantisation table (64 x uint16) index
    long long out_off;   // byte offset of the component's plane
    long long bw, bh;    // blocks per row / block rows (plane = bh*8 rows of bw*8 bytes)
    long long blk_start; // number of blocks of all planes before this one
};

struct JpegImage {       // int64 x 12
    long long H, W, ncomp, mode;      // mode 0: no subsampling, 1: h2v1 fancy, 2: h2v2 fancy, 3 / 4: h2v1 / h2v2 replicated
    long long y_off, y_stride, cb_off, cr_off, c_stride, ch, cw;      // chroma plane: ch x cw REAL samples
    long long out_off;                // byte offset of the H x W x ncomp output image
};

constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
              F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;

// one 8-point pass of jpeg_idct_islow (jidctint.c), v -> v, results descaled by `shift`
template <int SHIFT>
__device__ __forceinline__ void idct8(int (&v)[8]) {
    int z2 = v[2], z3 = v[6];
    int z1 = (z2 + z3) * F0_541;
    int tmp2 = z1 + z3 * (-F1_847);
    int tmp3 = z1 + z2 * F0_765;
    z2 = v[0]; z3 = v[4];
    int tmp0 = (z2 + z3) << 13;
    int tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 = z3 * -F1_961 + z5; z4 = z4 * -F0_390 + z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    v[0] = (tmp10 + tmp3 + R) >> SHIFT; v[7] = (tmp10 - tmp3 + R) >> SHIFT;
    v[1] = (tmp11 + tmp2 + R) >> SHIFT; v[6] = (tmp11 - tmp2 + R) >> SHIFT;
    v[2] = (tmp12 + tmp1 + R) >> SHIFT; v[5] = (tmp12 - tmp1 + R) >> SHIFT;
    v[3] = (tmp13 + tmp0 + R) >> SHIFT; v[4] = (tmp13 - tmp0 + R) >> SHIFT;
}

// libjpeg's range-limit table behind the IDCT (jdmaster.c prepare_range_limit_table), index masked to 10 bits
__device__ __forceinline__ unsigned range_limit(int x) {
    const int i = x & 1023;
    return i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896;
}

// 8 lanes per 8x8 block (lane j: column j in pass 1, row j in pass 2; the 8x8 transposition between the passes through LDS),
// 32 blocks per workgroup.
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const short* __restrict__ coef, const unsigned short* __restrict__ qt,
                                                        const JpegPlane* __restrict__ planes, int n_planes, long long total_blocks,
                                                        unsigned char* __restrict__ out) {
    __shared__ int ws[32][8][9];
    const int j = threadIdx.x & 7, lb = threadIdx.x >> 3;
    const long long gb = (long long)blockIdx.x * 32 + lb;
    const bool on = gb < total_blocks;
    int pl = 0;
    if (on) {        // the plane this block belongs to: last plane whose blk_start <= gb
        int lo = 0, hi = n_planes - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (planes[mid].blk_start <= gb) lo = mid; else hi = mid - 1;
        }
        pl = lo;
    }
    const JpegPlane P = planes[pl];
    const long long local = on ? gb - P.blk_start : 0;
    int v[8];
    if (on) {
        const short* c = coef + (P.coef_blk + local) * 64;
        const unsigned short* q = qt + P.qt_idx * 64;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (int)c[r * 8 + j] * (int)q[r * 8 + j];
        idct8<13 - 2>(v);                    // CONST_BITS - PASS1_BITS
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r][j] = v[r];
    }
    __syncthreads();
    if (on) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ws[lb][j][r];
        idct8<13 + 2 + 3>(v);                // CONST_BITS + PASS1_BITS + 3
        const unsigned lo4 = range_limit(v[0]) | (range_limit(v[1]) << 8) | (range_limit(v[2]) << 16) | (range_limit(v[3]) << 24);
        const unsigned hi4 = range_limit(v[4]) | (range_limit(v[5]) << 8) | (range_limit(v[6]) << 16) | (range_limit(v[7]) << 24);
        const long long by = local / P.bw, bx = local - by * P.bw;
        uint2* dst = reinterpret_cast<uint2*>(out + P.out_off + ((by * 8 + j) * P.bw + bx) * 8);
        *dst = make_uint2(lo4, hi4);
    }
}

__device__ __forceinline__ int clamp255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }

// one thread per output pixel; blockIdx.y = image
__global__ __launch_bounds__(256) void jpeg_to_rgb_kernel(const unsigned char* __restrict__ planes, const JpegImage* __restrict__ imgs,
                                                          unsigned char* __restrict__ out) {
    const JpegImage I = imgs[blockIdx.y];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I.H * I.W) return;
    const int y = (int)(idx / I.W), x = (int)(idx - (long long)y * I.W);
    const int Y = planes[I.y_off + (long long)y * I.y_stride + x];
    if (I.ncomp == 1) {
        out[I.out_off + idx] = (unsigned char)Y;
        return;
    }
    int cb, cr;
    if (I.mode == 0) {
        cb = planes[I.cb_off + (long long)y * I.c_stride + x];
        cr = planes[I.cr_off + (long long)y * I.c_stride + x];
    } else if (I.mode >= 3) {       // chroma planes of one or two columns: libjpeg replicates (h2v1_upsample / h2v2_upsample)
        const long long row = (I.mode == 4) ? (y >> 1) : y;
        cb = planes[I.cb_off + row * I.c_stride + (x >> 1)];
        cr = planes[I.cr_off + row * I.c_stride + (x >> 1)];
    } else {
        const int cw = (int)I.cw, ch = (int)I.ch;
        const int cx = x >> 1;
        const int xn = (x & 1) ? min(cx + 1, cw - 1) : max(cx - 1, 0);       // the horizontal neighbour (jdsample.c)
        const bool first = x == 0, last = x == 2 * cw - 1;
        if (I.mode == 1) {          // h2v1: (3 a + neighbour + {1 | 2}) >> 2, the edge columns copied
            const unsigned char* rb = planes + I.cb_off + (long long)y * I.c_stride;
            const unsigned char* rr = planes + I.cr_off + (long long)y * I.c_stride;
            const int bias = (x & 1) ? 2 : 1;
            cb = (first || last) ? rb[cx] : (3 * rb[cx] + rb[xn] + bias) >> 2;
            cr = (first || last) ? rr[cx] : (3 * rr[cx] + rr[xn] + bias) >> 2;
        } else {                    // h2v2: column sums 3 near + far over the two source rows, then (3 this + neighbour + {8 | 7}) >> 4
            const int cy = y >> 1;
            const int yn = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);   // rows beyond the plane: the edge row again
            const unsigned char* b0 = planes + I.cb_off + (long long)cy * I.c_stride;
            const unsigned char* b1 = planes + I.cb_off + (long long)yn * I.c_stride;
            const unsigned char* r0 = planes + I.cr_off + (long long)cy * I.c_stride;
            const unsigned char* r1 = planes + I.cr_off + (long long)yn * I.c_stride;
            const int bias = (x & 1) ? 7 : 8;
            const int tb = 3 * b0[cx] + b1[cx], nb = 3 * b0[xn] + b1[xn];
            const int tr = 3 * r0[cx] + r1[cx], nr = 3 * r0[xn] + r1[xn];
            cb = (first || last) ? (4 * tb + bias) >> 4 : (3 * tb + nb + bias) >> 4;
            cr = (first || last) ? (4 * tr + bias) >> 4 : (3 * tr + nr + bias) >> 4;
        }
    }
    cb -= 128; cr -= 128;
    const int r = clamp255(Y + ((91881 * cr + 32768) >> 16));
    const int g = clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    const int b = clamp255(Y + ((116130 * cb + 32768) >> 16));
    unsigned char* o = out + I.out_off + idx * 3;
    o[0] = (unsigned char)r; o[1] = (unsigned char)g; o[2] = (unsigned char)b;
}

}  // namespace

extern "C" {

// coef: DEVICE int16 [total_blocks][64] (natural order, as witw_jpeg_decode_coef writes them); qt: DEVICE uint16 [n_tables][64];
// planes: DEVICE int64 [n_planes][6] = {first block, table index, byte offset of the output plane, blocks wide, blocks high,
// blocks of all earlier planes}; out: DEVICE bytes, the 8-bit component planes (bh*8 rows of bw*8 bytes each).
int witw_jpeg_idct(const void* coef, const void* qt, const void* planes, int n_planes, long long total_blocks, void* out, void* stream) {
    WITW_CHECK_ARG(coef && qt && planes && out, "jpeg_idct: null pointer");
    WITW_CHECK_ARG(n_planes > 0 && total_blocks > 0 && total_blocks < (1LL << 36), "jpeg_idct: %d planes, %lld blocks", n_planes, total_blocks);
    const unsigned grid = (unsigned)((total_blocks + 31) / 32);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const short*)coef, (const unsigned short*)qt,
                       (const JpegPlane*)planes, n_planes, total_blocks, (unsigned char*)out);
    WITW_CHECK_LAUNCH("jpeg_idct");
    return WITW_OK;
}

// planes: the output of witw_jpeg_idct; images: DEVICE int64 [n_images][12] = {H, W, components (1 | 3), mode (0: chroma at full
// size, 1: h2v1, 2: h2v2 fancy upsampling, 3 / 4: the same by replication -- libjpeg's choice for chroma planes of <= 2 columns), luma plane offset, luma stride, Cb offset, Cr offset, chroma stride, chroma rows, chroma columns (real
// samples: ceil(H / 2) ...), output offset}; max_pixels = the largest H*W; out: H x W x components interleaved bytes per image.
int witw_jpeg_to_rgb(const void* planes, const void* images, int n_images, long long max_pixels, void* out, void* stream) {
    WITW_CHECK_ARG(planes && images && out, "jpeg_to_rgb: null pointer");
    WITW_CHECK_ARG(n_images > 0 && n_images <= 65535 && max_pixels > 0 && max_pixels < (1LL << 38), "jpeg_to_rgb: %d images, %lld pixels", n_images, max_pixels);
    hipLaunchKernelGGL(jpeg_to_rgb_kernel, dim3((unsigned)((max_pixels + 255) / 256), (unsigned)n_images), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)planes, (const JpegImage*)images, (unsigned char*)out);
    WITW_CHECK_LAUNCH("jpeg_to_rgb");
    return WITW_OK;
}

}  // extern "C"
