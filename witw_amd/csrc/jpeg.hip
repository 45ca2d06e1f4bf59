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

// ---- entropy decoding on the device (round 6; SURVEY 8(f)3 'decode ... entirely on device'), step one: files that carry RESTART
// MARKERS. A restart interval starts byte-aligned with the DC predictors reset, so intervals decode independently: ONE THREAD PER
// INTERVAL (a 512 x 512 4:2:0 file written with a marker per MCU row has 32 of them, a batch of 128 pairs ~5,900), one wave per
// file. The host's share shrinks to a byte scan for the markers (csrc_host/jpeg_coef.cpp, witw_jpeg_entropy_plan: the plan layout is
// described there) and the FILE BYTES cross PCIe instead of the coefficient blocks (~80 KB instead of 786 KB per overhead image).
// Integer / byte work, latency-bound per thread (a table look-up and a few shifts per symbol); nothing here is MFMA- or
// HBM-shaped. Same coefficients, bit for bit, as witw_jpeg_decode_coef (tests/test_jpeg_gpu.py).
struct HuffLds {
    unsigned short look[512];      // 9-bit prefix -> (code length << 8) | symbol; 0: the code is longer than 9 bits
    int maxcode[18];               // largest code of each length (-1: none), [17] = sentinel
    int valoff[17];                // symbol index of the first code of a length minus that code
    unsigned char sym[256];
};

struct JpegFileDev {               // int64 x 4 per file
    long long bytes;               // address of the file bytes (8-byte aligned, at least 24 readable bytes behind the end)
    long long plan;                // address of the plan (witw_jpeg_entropy_plan)
    long long coef;                // address of the file's coefficient area: int16 [blocks][64], ZERO-FILLED by the caller
    long long n_bytes;             // file length
};

__constant__ unsigned char kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                          41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                          30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct BitReader {                 // over the STUFFED bytes of one restart interval: FF 00 -> FF on the fly, any other FF xx ends the data
    const unsigned long long* words;
    unsigned pos, end;             // byte offsets in the file
    unsigned long long cache;      // the aligned 8 bytes that hold byte `pos`
    unsigned long long ahead;      // ... and the 8 bytes behind them, requested when `cache` was taken (the stream is read in order:
                                   // the load's round trip runs under the decoding of the current word instead of in front of the next)
    unsigned cidx;                 // which 8-byte word `cache` is (0xffffffff: none)
    unsigned long long buf;        // bits, left-aligned
    int n;                         // valid bits in buf
    int starved;                   // zero bytes fed behind the end of the data

    __device__ __forceinline__ unsigned raw(unsigned p) {
        if ((p >> 3) != cidx) {
            const unsigned w = p >> 3;
            cache = (w == cidx + 1u) ? ahead : words[w];
            cidx = w;
            ahead = words[w + 1u];
        }
        return (unsigned)(cache >> (8 * (p & 7))) & 0xffu;
    }
    __device__ __forceinline__ void fill() {      // tops up to more than 56 valid bits
        while (n <= 56) {
            unsigned b = 0;
            if (pos < end) {
                b = raw(pos);
                if (b == 0xffu) {
                    const unsigned b2 = pos + 1 < end ? raw(pos + 1) : 0xd9u;
                    if (b2 == 0u) pos += 2;                    // a stuffed FF
                    else { end = pos; b = 0; ++starved; }      // a marker (the next interval's RSTn, or EOI): the data ends here
                } else {
                    ++pos;
                }
            } else {
                ++starved;
            }
            buf |= (unsigned long long)b << (56 - n);
            n += 8;
        }
    }
    __device__ __forceinline__ int get(int k) {    // k in 1..16, caller has >= 32 valid bits
        const int v = (int)(buf >> (64 - k));
        buf <<= k;
        n -= k;
        return v;
    }
};

__device__ __forceinline__ int huff_decode(BitReader& b, const HuffLds& h) {      // caller has >= 32 valid bits; -1: invalid code
    const unsigned e = h.look[(unsigned)(b.buf >> 55)];
    if (e) {
        const int len = (int)(e >> 8);
        b.buf <<= len;
        b.n -= len;
        return (int)(e & 255u);
    }
    int len = 10;
    int code = (int)(b.buf >> 54);
    while (code > h.maxcode[len]) {
        ++len;
        if (len > 16) return -1;
        code = (int)(b.buf >> (64 - len));
    }
    b.buf <<= len;
    b.n -= len;
    return h.sym[(code + h.valoff[len]) & 255];
}

__device__ __forceinline__ int jpeg_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const JpegFileDev* __restrict__ files, int* __restrict__ errors) {
    __shared__ HuffLds tab[4];                    // DC slot 0, 1, AC slot 0, 1
    __shared__ int hdr[32];
    const int lane = threadIdx.x;
    const JpegFileDev f = files[blockIdx.x];
    const unsigned char* plan = reinterpret_cast<const unsigned char*>(f.plan);
    // blockIdx.y: which 64 intervals of the file this wave decodes (a thread's chain of symbols is what bounds the kernel: files with
    // short intervals spread over several waves); waves past the file's last interval leave before building tables
    if ((int)blockIdx.y * 64 >= reinterpret_cast<const int*>(plan)[1] && reinterpret_cast<const int*>(plan)[0] == 0x3157504A) return;
    if (lane < 32) hdr[lane] = reinterpret_cast<const int*>(plan)[lane];
    // ---- the four decoding tables from the DHT counts / symbols: lanes 0-3 assign the canonical codes of one table each (16
    // lengths), then all lanes fill the 9-bit look-ups
    if (lane < 4) {
        const unsigned char* d = lane < 2 ? plan + 128 + 32 * lane : plan + 192 + 272 * (lane - 2);
        HuffLds& h = tab[lane];
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            h.valoff[len] = k - code;
            const int cnt = d[len - 1];
            k += cnt;
            code += cnt;
            h.maxcode[len] = cnt ? code - 1 : -1;
            code <<= 1;
        }
        h.maxcode[17] = 0x7fffffff;
        h.maxcode[0] = -1;
        h.valoff[0] = 0;
        const int nsym = lane < 2 ? (k < 16 ? k : 16) : (k < 256 ? k : 256);
        for (int i = 0; i < 256; ++i) h.sym[i] = i < nsym ? d[16 + i] : 0;
    }
    __syncthreads();
    for (int e = lane; e < 4 * 512; e += 64) {
        const int t = e >> 9, i = e & 511;
        const unsigned char* d = t < 2 ? plan + 128 + 32 * t : plan + 192 + 272 * (t - 2);
        const HuffLds& h = tab[t];
        unsigned short v = 0;
        for (int len = 1; len <= 9; ++len) {
            const int code = i >> (9 - len);
            const int cnt = d[len - 1];
            if (cnt && code <= h.maxcode[len] && code > h.maxcode[len] - cnt) {
                v = (unsigned short)((len << 8) | h.sym[(code + h.valoff[len]) & 255]);
                break;
            }
        }
        tab[t].look[i] = v;
    }
    __syncthreads();
    if (hdr[0] != 0x3157504A) {
        if (lane == 0 && blockIdx.y == 0) errors[blockIdx.x] = 2;
        return;
    }
    const int n_int = hdr[1], restart = hdr[2], mcux = hdr[3], mcuy = hdr[4], ncomp = hdr[5];
    const unsigned end_all = (unsigned)hdr[27];
    const unsigned* ioff = reinterpret_cast<const unsigned*>(plan + 736);
    short* coef = reinterpret_cast<short*>(f.coef);
    const long long mcus = (long long)mcux * mcuy;
    bool bad = false;
    for (int iv = (int)blockIdx.y * 64 + lane; iv < n_int; iv += 64 * (int)gridDim.y) {
        BitReader b;
        b.words = reinterpret_cast<const unsigned long long*>(f.bytes);
        b.pos = ioff[iv];
        b.end = iv + 1 < n_int ? ioff[iv + 1] : end_all;      // (the RSTn marker in front of the next interval stops the reader earlier)
        if (b.end > (unsigned)f.n_bytes) b.end = (unsigned)f.n_bytes;
        b.cidx = 0xfffffff0u; b.cache = 0; b.ahead = 0; b.buf = 0; b.n = 0; b.starved = 0;
        int pred[3] = {0, 0, 0};
        const long long m0 = (long long)iv * restart;
        const long long m1 = m0 + restart < mcus ? m0 + restart : mcus;
        for (long long m = m0; m < m1 && !bad; ++m) {
            const int my = (int)(m / mcux), mx = (int)(m - (long long)my * mcux);
            for (int k = 0; k < ncomp && !bad; ++k) {
                const int c = hdr[28 + k];
                const int* q = hdr + 6 + 7 * c;
                const int ch = q[0], cv = q[1], cbw = q[2];
                const long long coff = q[4];
                const HuffLds& hd = tab[q[5] & 1];
                const HuffLds& ha = tab[2 + (q[6] & 1)];
                for (int v = 0; v < cv && !bad; ++v)
                    for (int hh = 0; hh < ch && !bad; ++hh) {
                        short* blk = coef + (coff + (long long)(my * cv + v) * cbw + (mx * ch + hh)) * 64;
                        b.fill();
                        int s = huff_decode(b, hd);
                        if (s < 0 || s > 15) { bad = true; break; }
                        if (s) pred[c] += jpeg_extend(b.get(s), s);
                        blk[0] = (short)pred[c];
                        for (int kk = 1; kk < 64;) {
                            if (b.n < 32) b.fill();
                            const int rs = huff_decode(b, ha);
                            if (rs < 0) { bad = true; break; }
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                kk += 16;
                                continue;
                            }
                            kk += r;
                            if (kk > 63) { bad = true; break; }
                            blk[kZigZag[kk]] = (short)jpeg_extend(b.get(s), s);
                            ++kk;
                        }
                        if (b.starved > 9) bad = true;      // the data ended inside the interval (witw_jpeg_decode_coef: -3)
                    }
            }
        }
    }
    if (bad) errors[blockIdx.x] = 1;
}

}  // namespace

extern "C" {

// Entropy decoding of n_files JPEG files WITH RESTART MARKERS on the device, one thread per restart interval: files = DEVICE int64
// [n_files][4] = {address of the file bytes (8-byte aligned, 24 readable bytes behind the end), address of the file's plan
// (witw_jpeg_entropy_plan, 4-byte aligned), address of its coefficient area int16 [blocks][64] (zero-filled by the caller), file
// length}; max_intervals = the largest number of restart intervals of a file of the launch (sizes the grid: one wave per 64
// intervals of a file); errors = DEVICE int32 [n_files], zeroed by the caller: 1 where the entropy-coded data of a file is damaged (its
// coefficients are then incomplete: witw_jpeg_decode_coef returns -3 for such a file), 2 for a bad plan. The coefficients are the
// bits witw_jpeg_decode_coef writes; witw_jpeg_idct / witw_jpeg_to_rgb take it from there (model/cvig_fov.py:88-89).
int witw_jpeg_huffman(const void* files, int n_files, int max_intervals, int* errors, void* stream) {
    WITW_CHECK_ARG(files && errors, "jpeg_huffman: null pointer");
    WITW_CHECK_ARG(n_files > 0 && max_intervals > 0, "jpeg_huffman: %d files, %d intervals", n_files, max_intervals);
    int waves = (max_intervals + 63) / 64;      // per file; at most 64 (a file with more than 4096 intervals loops)
    if (waves > 64) waves = 64;
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3((unsigned)n_files, (unsigned)waves), dim3(64), 0, (hipStream_t)stream,
                       (const JpegFileDev*)files, errors);
    WITW_CHECK_LAUNCH("jpeg_huffman");
    return WITW_OK;
}

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
