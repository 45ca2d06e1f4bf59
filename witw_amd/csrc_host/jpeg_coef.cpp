// Host side of the on-device JPEG decode (the reference decodes on the host inside its DataLoader workers:
// skimage.io.imread -> PIL -> libjpeg, model/cvig_fov.py:88-89, :402): ENTROPY decoding only -- marker parsing, Huffman decoding,
// DC prediction, de-zigzag -- of baseline / extended-sequential 8-bit JFIF files into quantised DCT coefficient blocks. Everything
// behind it (dequantisation, the integer 'islow' inverse DCT, chroma upsampling, YCbCr -> RGB) runs on the GPU
// (csrc/jpeg.hip). Plain C ABI, no GPU runtime: DataLoader workers load this library, never libwitw_hip.so.
//
// Coefficient layout: component c holds bh[c] x bw[c] blocks (the MCU-padded block grid, raster order), each 64 int16 in NATURAL
// (row-major) order; components follow one another. Quantisation tables: 64 uint16 per component, natural order.
#include <stdint.h>
#include <stddef.h>
#include <string.h>

namespace {

const uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int FAST = 10;          // look-ahead bits of the decoding tables

struct Huff {
    uint8_t look_len[1 << FAST];     // FAST-bit prefix -> code length (0: longer than FAST bits)
    uint8_t look_sym[1 << FAST];
    int32_t fast_ac[1 << FAST];      // AC tables: (value << 16) | (run << 8) | (code length + magnitude bits) when code AND magnitude
                                     // bits fit the look-ahead, else 0 (the usual two-step path)
    int32_t maxcode[18];             // largest code of each length (-1: none), [17] = sentinel
    int32_t valoff[17];              // symbol index of the first code of each length minus that code
    uint8_t sym[256];
    uint8_t counts[16];              // codes per length as the DHT segment gives them (witw_jpeg_entropy_plan hands them to the device)
    int nsym;
    bool present;
};

bool build_huff(Huff& h, const uint8_t* counts, const uint8_t* symbols, int nsym, bool ac) {
    memcpy(h.sym, symbols, nsym);
    memcpy(h.counts, counts, 16);
    h.nsym = nsym;
    memset(h.look_len, 0, sizeof(h.look_len));
    memset(h.fast_ac, 0, sizeof(h.fast_ac));
    h.present = false;
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        h.valoff[len] = k - code;
        // an over-subscribed length (more codes than the code space has left) is rejected BEFORE anything is written: with
        // code + count <= 2^len every look-up index below stays under 2^FAST and k stays under nsym <= 256
        if (code + counts[len - 1] > (1 << len) || k + counts[len - 1] > nsym) return false;
        for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
            if (len <= FAST) {
                const int first = code << (FAST - len), n = 1 << (FAST - len);
                for (int j = 0; j < n; ++j) {
                    h.look_len[first + j] = (uint8_t)len;
                    h.look_sym[first + j] = symbols[k];
                }
            }
        }
        h.maxcode[len] = counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    if (k != nsym) return false;
    h.present = true;
    if (ac)
        for (int i = 0; i < (1 << FAST); ++i) {
            const int len = h.look_len[i];
            if (!len) continue;
            const int rs = h.look_sym[i], run = rs >> 4, mag = rs & 15;
            if (mag == 0 || len + mag > FAST) continue;
            int v = (i >> (FAST - len - mag)) & ((1 << mag) - 1);       // the magnitude bits behind the code
            if (v < (1 << (mag - 1))) v -= (1 << mag) - 1;
            h.fast_ac[i] = (int32_t)((uint32_t)v << 16) | (run << 8) | (len + mag);
        }
    return true;
}

// Bit reader over the UNSTUFFED entropy-coded bytes (FF 00 -> FF done once up front; 8 zero bytes behind the end)
struct Bits {
    const uint8_t* p;
    uint64_t buf;       // bits left-aligned
    int n;              // valid bits in buf
    void init(const uint8_t* a) { p = a; buf = 0; n = 0; }
    inline void fill() {      // tops up to at least 56 valid bits: one unaligned 8-byte load, whole bytes appended
        uint64_t w;
        memcpy(&w, p, 8);
        buf |= __builtin_bswap64(w) >> n;
        const int adv = (63 - n) >> 3;
        p += adv;
        n += adv << 3;
    }
    inline int peek(int k) const { return (int)(buf >> (64 - k)); }
    inline void skip(int k) { buf <<= k; n -= k; }
    inline int get(int k) { const int v = peek(k); skip(k); return v; }
};

inline int decode_sym(Bits& b, const Huff& h) {      // caller has >= 16 valid bits
    const int look = b.peek(FAST);
    int len = h.look_len[look];
    if (len) { b.skip(len); return h.look_sym[look]; }
    len = FAST + 1;
    int code = b.peek(len);
    while (code > h.maxcode[len]) { ++len; if (len > 16) return -1; code = b.peek(len); }
    b.skip(len);
    return h.sym[(code + h.valoff[len]) & 255];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Comp { int id, h, v, tq, td, ta, bw, bh; int64_t off; int pred; };

struct Parsed {
    int H, W, ncomp, hmax, vmax, mcux, mcuy, restart;
    Comp c[4];
    uint16_t qt[4][64];
    bool qt_present[4];
    Huff dc[4], ac[4];
    const uint8_t* scan;      // start of the entropy-coded segment
    int scan_ncomp, scan_comp[4];
    bool jfif, adobe;         // APP0 'JFIF' / APP14 'Adobe' seen (libjpeg picks the colour space from them, jdapimin.c)
    int adobe_transform;
    int status;               // 0 ok, < 0 see witw_jpeg_* below
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

int parse(const uint8_t* d, size_t n, Parsed& P) {
    memset(&P, 0, sizeof(P));
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return -1;
    size_t i = 2;
    bool have_sof = false;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) return -1;
        while (i < n && d[i] == 0xFF) ++i;      // fill bytes
        if (i >= n) return -1;
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -1;               // EOI before a scan
        if (i + 2 > n) return -1;
        const int len = be16(d + i);
        if (len < 2 || i + len > n) return -1;
        const uint8_t* s = d + i + 2;
        const int sl = len - 2;
        if (m == 0xDB) {
            int k = 0;
            while (k < sl) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                ++k;
                if (tq > 3 || pq > 1 || k + 64 * (pq + 1) > sl) return -1;
                for (int j = 0; j < 64; ++j) {
                    P.qt[tq][ZZ[j]] = (uint16_t)(pq ? be16(s + k + 2 * j) : s[k + j]);
                }
                P.qt_present[tq] = true;
                k += 64 * (pq + 1);
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return -2;                   // 8-bit samples only
            P.H = be16(s + 1); P.W = be16(s + 3); P.ncomp = s[5];
            if (P.H <= 0 || P.W <= 0 || (P.ncomp != 1 && P.ncomp != 3) || sl < 6 + 3 * P.ncomp) return -2;
            for (int c = 0; c < P.ncomp; ++c) {
                P.c[c].id = s[6 + 3 * c];
                P.c[c].h = s[7 + 3 * c] >> 4; P.c[c].v = s[7 + 3 * c] & 15;
                P.c[c].tq = s[8 + 3 * c];
                if (P.c[c].h < 1 || P.c[c].h > 2 || P.c[c].v < 1 || P.c[c].v > 2 || P.c[c].tq > 3) return -2;
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return -2;                                              // progressive / lossless / arithmetic: host decoder
        } else if (m == 0xC4) {
            int k = 0;
            while (k < sl) {
                if (k + 17 > sl) return -1;
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) return -1;
                int nsym = 0;
                for (int j = 0; j < 16; ++j) nsym += s[k + 1 + j];
                if (nsym > 256 || k + 17 + nsym > sl) return -1;
                if (!build_huff(tc ? P.ac[th] : P.dc[th], s + k + 1, s + k + 17, nsym, tc != 0)) return -1;
                k += 17 + nsym;
            }
        } else if (m == 0xE0) {                                     // jdmarker.c examine_app0
            if (sl >= 14 && !memcmp(s, "JFIF\0", 5)) P.jfif = true;
        } else if (m == 0xEE) {                                     // jdmarker.c examine_app14
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) { P.adobe = true; P.adobe_transform = s[11]; }
        } else if (m == 0xDD) {
            if (sl < 2) return -1;
            P.restart = be16(s);
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1) return -1;
            const int ns = s[0];
            if (ns != P.ncomp || sl < 1 + 2 * ns + 3) return -2;   // one interleaved scan with every component
            for (int k = 0; k < ns; ++k) {
                int ci = -1;
                for (int c = 0; c < P.ncomp; ++c) if (P.c[c].id == s[1 + 2 * k]) ci = c;
                if (ci < 0) return -1;
                P.scan_comp[k] = ci;
                P.c[ci].td = s[2 + 2 * k] >> 4; P.c[ci].ta = s[2 + 2 * k] & 15;
                if (P.c[ci].td > 3 || P.c[ci].ta > 3) return -1;
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return -2;
            P.scan_ncomp = ns;
            P.scan = d + i + len;
            break;
        }
        i += len;
    }
    if (!P.scan) return -1;
    P.hmax = P.vmax = 1;
    for (int c = 0; c < P.ncomp; ++c) { if (P.c[c].h > P.hmax) P.hmax = P.c[c].h; if (P.c[c].v > P.vmax) P.vmax = P.c[c].v; }
    if (P.ncomp == 1) { P.c[0].h = P.c[0].v = 1; P.hmax = P.vmax = 1; }      // a single-component scan is never interleaved
    else if (P.c[1].h != 1 || P.c[1].v != 1 || P.c[2].h != 1 || P.c[2].v != 1 || P.c[0].h != P.hmax || P.c[0].v != P.vmax) return -2;
    // the device back end upsamples h1v1, h2v1 and h2v2 only: 4:4:0 (h1v2, e.g. a losslessly rotated 4:2:2 file) goes to the host decoder
    if (P.hmax == 1 && P.vmax == 2) return -2;
    // colour space as libjpeg's default_decompress_parms (jdapimin.c) decides it for three components: JFIF => YCbCr; else an
    // Adobe marker's transform flag (0 = RGB, 1 = YCbCr); else the component ids ('R','G','B' = RGB). The device back end always
    // applies YCbCr -> RGB, so every other case (and an unknown transform, which libjpeg warns about) is the host decoder's.
    if (P.ncomp == 3 && !P.jfif) {
        if (P.adobe) { if (P.adobe_transform != 1) return -2; }
        else if (P.c[0].id == 'R' && P.c[1].id == 'G' && P.c[2].id == 'B') return -2;
    }
    P.mcux = (P.W + 8 * P.hmax - 1) / (8 * P.hmax);
    P.mcuy = (P.H + 8 * P.vmax - 1) / (8 * P.vmax);
    int64_t off = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        P.c[c].bw = P.mcux * P.c[c].h; P.c[c].bh = P.mcuy * P.c[c].v;
        P.c[c].off = off;
        off += (int64_t)P.c[c].bw * P.c[c].bh;
        if (!P.qt_present[P.c[c].tq] || !P.dc[P.c[c].td].present || !P.ac[P.c[c].ta].present) return -1;
    }
    return 0;
}

// one Parsed per thread, on the HEAP (not in the TLS block): a sanitizer build red-zones it, so an overrun of the tables is seen
Parsed& parsed() {
    static thread_local Parsed* p = nullptr;
    if (!p) p = new Parsed;
    return *p;
}

}  // namespace

extern "C" {

// info[0..]: H, W, ncomp, hmax, vmax, total blocks, then per component (4 slots): h, v, blocks wide, blocks high.
// Returns 0; -1 = not a decodable JPEG stream; -2 = a JPEG this decoder leaves to the host library (progressive, arithmetic,
// 12-bit, CMYK, non-interleaved scans, sampling other than 4:4:4 / 4:2:2 (h2v1) / 4:2:0 / grey, RGB-colourspace files: Adobe
// transform 0 or component ids 'R','G','B' without a JFIF marker).
int witw_jpeg_info(const uint8_t* data, size_t n, int32_t* info /* 22 */) {
    Parsed& P = parsed();
    const int rc = parse(data, n, P);
    if (rc) return rc;
    info[0] = P.H; info[1] = P.W; info[2] = P.ncomp; info[3] = P.hmax; info[4] = P.vmax;
    int64_t total = 0;
    for (int c = 0; c < 4; ++c) {
        const bool on = c < P.ncomp;
        info[6 + 4 * c] = on ? P.c[c].h : 0; info[7 + 4 * c] = on ? P.c[c].v : 0;
        info[8 + 4 * c] = on ? P.c[c].bw : 0; info[9 + 4 * c] = on ? P.c[c].bh : 0;
        if (on) total += (int64_t)P.c[c].bw * P.c[c].bh;
    }
    info[5] = (int32_t)total;
    return 0;
}

// coef: total blocks x 64 int16 (zero-filled here); qt: ncomp x 64 uint16. Returns 0 or a negative code as above; -3 = the
// entropy-coded data ended early or holds an invalid code (the blocks decoded so far are kept, the rest stay zero).
int witw_jpeg_decode_coef(const uint8_t* data, size_t n, int16_t* coef, uint16_t* qt) {
    Parsed& P = parsed();
    int rc = parse(data, n, P);
    if (rc) return rc;
    int64_t total = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        total += (int64_t)P.c[c].bw * P.c[c].bh;
        memcpy(qt + 64 * c, P.qt[P.c[c].tq], 128);
        P.c[c].pred = 0;
    }
    memset(coef, 0, (size_t)total * 128);
    // ---- unstuff the entropy-coded segment once: FF 00 -> FF, RSTn markers dropped (their byte positions kept), any other
    // marker ends the data; zeros behind the end so that the bit reader may run ahead
    static thread_local uint8_t* clean = nullptr;
    static thread_local size_t clean_cap = 0;
    static thread_local uint32_t* rst_pos = nullptr;
    static thread_local size_t rst_cap = 0;
    const size_t seg = (size_t)(data + n - P.scan);
    // PAD zero bytes behind the data: a block of a corrupt stream can read at most 64 x (16 + 15) bits = 248 bytes past the end
    // before the per-block check below stops the decode
    constexpr size_t PAD = 512;
    if (clean_cap < seg + PAD) { delete[] clean; clean_cap = seg + PAD + seg / 4; clean = new uint8_t[clean_cap]; }
    const size_t max_rst = P.restart ? (size_t)P.mcux * P.mcuy / P.restart + 2 : 1;
    if (rst_cap < max_rst) { delete[] rst_pos; rst_cap = max_rst + max_rst / 4; rst_pos = new uint32_t[rst_cap]; }
    size_t w = 0, n_rst = 0;
    {
        const uint8_t* q = P.scan;
        const uint8_t* e = data + n;
        while (q < e) {
            const uint8_t* f = (const uint8_t*)memchr(q, 0xFF, (size_t)(e - q));
            if (!f) { memcpy(clean + w, q, (size_t)(e - q)); w += (size_t)(e - q); break; }
            memcpy(clean + w, q, (size_t)(f - q));
            w += (size_t)(f - q);
            if (f + 1 >= e) break;
            const int m = f[1];
            if (m == 0) { clean[w++] = 0xFF; q = f + 2; }
            else if (m >= 0xD0 && m <= 0xD7) { if (n_rst < rst_cap) rst_pos[n_rst++] = (uint32_t)w | ((uint32_t)(m - 0xD0) << 29); q = f + 2; }
            else if (m == 0xFF) { q = f + 1; }
            else break;                                     // EOI or another marker: the entropy-coded data ends here
        }
    }
    memset(clean + w, 0, PAD);     // the reader loads 8 bytes at a time and may run ahead by one block
    Bits b;
    b.init(clean);
    int to_restart = P.restart, next_rst = 0;
    size_t rst_i = 0;
    const uint8_t* const limit = clean + w + 9;             // reading this far past the data means the stream was too short
    for (int my = 0; my < P.mcuy; ++my)
        for (int mx = 0; mx < P.mcux; ++mx) {
            if (P.restart && to_restart == 0) {
                if (rst_i >= n_rst || (int)(rst_pos[rst_i] >> 29) != next_rst) return -3;
                b.init(clean + (rst_pos[rst_i] & 0x1fffffffu));
                ++rst_i;
                next_rst = (next_rst + 1) & 7;
                to_restart = P.restart;
                for (int c = 0; c < P.ncomp; ++c) P.c[c].pred = 0;
            }
            for (int k = 0; k < P.scan_ncomp; ++k) {
                Comp& C = P.c[P.scan_comp[k]];
                const Huff& hd = P.dc[C.td];
                const Huff& ha = P.ac[C.ta];
                for (int v = 0; v < C.v; ++v)
                    for (int h = 0; h < C.h; ++h) {
                        int16_t* blk = coef + (C.off + (int64_t)(my * C.v + v) * C.bw + (mx * C.h + h)) * 64;
                        b.fill();                           // >= 56 bits: a code (<= 16) + its magnitude bits (<= 15), twice over
                        int s = decode_sym(b, hd);
                        if (s < 0 || s > 15) return -3;
                        if (s) C.pred += extend(b.get(s), s);
                        blk[0] = (int16_t)C.pred;
                        for (int kk = 1; kk < 64;) {
                            if (b.n < 32) b.fill();
                            const int32_t fa = ha.fast_ac[b.peek(FAST)];
                            if (fa) {                       // code and value in one look-up
                                kk += (fa >> 8) & 15;
                                if (kk > 63) return -3;
                                b.skip(fa & 255);
                                blk[ZZ[kk++]] = (int16_t)(fa >> 16);
                                continue;
                            }
                            const int rs = decode_sym(b, ha);
                            if (rs < 0) return -3;
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                kk += 16;
                                continue;
                            }
                            kk += r;
                            if (kk > 63) return -3;
                            blk[ZZ[kk]] = (int16_t)extend(b.get(s), s);
                            ++kk;
                        }
                        if (b.p > limit) return -3;
                    }
            }
            if (P.restart) --to_restart;
        }
    return 0;
}

// ---- entropy decoding ON THE DEVICE for files that carry restart markers (csrc/jpeg.hip, jpeg_huffman_kernel: one GPU thread per
// restart interval). What is left for the host is a byte scan: the header, and the positions of the RSTn markers in the entropy-coded
// segment (memchr over ~80 KB per 512 x 512 image instead of decoding ~60 k Huffman symbols). The plan written here is what the
// kernel reads beside the FILE BYTES themselves (it unstuffs FF 00 on the fly):
//   int32[0] magic 'JPW1', [1] intervals, [2] MCUs per interval, [3] MCUs per row, [4] MCU rows, [5] components,
//   [6 + 7c ..] component c < 3: h, v, blocks wide, blocks high, first block (within the file's coefficient area), DC slot, AC slot,
//   [27] end of the entropy-coded data (byte offset in the file), [28..30] component of scan position k, [31] 0;
//   byte 128: DC tables of slots 0, 1 (16 counts + 16 symbols each), byte 192: AC tables of slots 0, 1 (16 counts + 256 symbols each);
//   byte 736: uint32 [intervals]: byte offset in the file of each interval's first entropy-coded byte.
// A file without restart markers gets a plan of ONE interval (the whole scan): the device decodes it with the self-synchronising
// kernel. Files with more than two DC or AC tables in use, or whose markers are out of sequence, return -2 / -3 and take the host
// path (witw_jpeg_decode_coef) as before.
enum { WITW_JPEG_PLAN_FIXED = 736 };

long long witw_jpeg_entropy_plan_bytes(const uint8_t* data, size_t n) {      // 0: not a file for this decoder
    Parsed& P = parsed();
    if (parse(data, n, P)) return 0;
    if (P.restart <= 0) return WITW_JPEG_PLAN_FIXED + 4;      // no restart markers: ONE interval, decoded by the self-synchronising kernel
    const long long mcus = (long long)P.mcux * P.mcuy;
    return WITW_JPEG_PLAN_FIXED + 4 * ((mcus + P.restart - 1) / P.restart);
}

// plan: plan_cap bytes (witw_jpeg_entropy_plan_bytes); qt: ncomp x 64 uint16 (as witw_jpeg_decode_coef writes them).
// Returns the plan's size in bytes, or -1 / -2 as witw_jpeg_info, -2 also for files with more than two Huffman tables of a kind in
// use, -3 when the restart markers found do not number intervals - 1 in sequence.
long long witw_jpeg_entropy_plan(const uint8_t* data, size_t n, uint8_t* plan, size_t plan_cap, uint16_t* qt) {
    Parsed& P = parsed();
    const int rc = parse(data, n, P);
    if (rc) return rc;
    // no restart markers: the whole scan is ONE interval ([2] = every MCU of the image); the device finds its way into the middle of
    // it by self-synchronisation (csrc/jpeg.hip, jpeg_selfsync_kernel)
    const long long mcus = (long long)P.mcux * P.mcuy;
    if (P.restart <= 0) {
        if (mcus > 0x7fffffff) return -2;
        P.restart = (int)mcus;
    }
    const long long n_int = (mcus + P.restart - 1) / P.restart;
    const long long need = WITW_JPEG_PLAN_FIXED + 4 * n_int;
    if ((long long)plan_cap < need || n_int > 0x7fffffff) return -1;
    memset(plan, 0, (size_t)need);
    int32_t* h = reinterpret_cast<int32_t*>(plan);
    h[0] = 0x3157504A;      // 'JPW1'
    h[1] = (int32_t)n_int; h[2] = P.restart; h[3] = P.mcux; h[4] = P.mcuy; h[5] = P.ncomp;
    int dc_slot[4] = {-1, -1, -1, -1}, ac_slot[4] = {-1, -1, -1, -1}, n_dc = 0, n_ac = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        if (dc_slot[P.c[c].td] < 0) { if (n_dc == 2) return -2; dc_slot[P.c[c].td] = n_dc++; }
        if (ac_slot[P.c[c].ta] < 0) { if (n_ac == 2) return -2; ac_slot[P.c[c].ta] = n_ac++; }
        int32_t* q = h + 6 + 7 * c;
        q[0] = P.c[c].h; q[1] = P.c[c].v; q[2] = P.c[c].bw; q[3] = P.c[c].bh; q[4] = (int32_t)P.c[c].off;
        q[5] = dc_slot[P.c[c].td]; q[6] = ac_slot[P.c[c].ta];
        memcpy(qt + 64 * c, P.qt[P.c[c].tq], 128);
    }
    for (int k = 0; k < P.scan_ncomp && k < 3; ++k) h[28 + k] = P.scan_comp[k];
    for (int t = 0; t < 4; ++t) {
        if (dc_slot[t] >= 0) {
            const Huff& hf = P.dc[t];
            if (hf.nsym > 16) return -2;                              // a DC table holds at most 12 categories (16 leaves room)
            uint8_t* d = plan + 128 + 32 * dc_slot[t];
            memcpy(d, hf.counts, 16);
            memcpy(d + 16, hf.sym, (size_t)hf.nsym);
        }
        if (ac_slot[t] >= 0) {
            const Huff& hf = P.ac[t];
            uint8_t* d = plan + 192 + 272 * ac_slot[t];
            memcpy(d, hf.counts, 16);
            memcpy(d + 16, hf.sym, (size_t)hf.nsym);
        }
    }
    uint32_t* off = reinterpret_cast<uint32_t*>(plan + WITW_JPEG_PLAN_FIXED);
    const uint8_t* q = P.scan;
    const uint8_t* e = data + n;
    long long found = 0;
    off[0] = (uint32_t)(P.scan - data);
    int next_rst = 0;
    const uint8_t* end = e;
    while (q < e) {
        const uint8_t* f = (const uint8_t*)memchr(q, 0xFF, (size_t)(e - q));
        if (!f || f + 1 >= e) break;
        const int m = f[1];
        if (m == 0) { q = f + 2; continue; }                           // a stuffed FF
        if (m == 0xFF) { q = f + 1; continue; }                        // fill byte
        if (m >= 0xD0 && m <= 0xD7) {
            if (m - 0xD0 != next_rst || found + 1 >= n_int) return -3;
            next_rst = (next_rst + 1) & 7;
            off[++found] = (uint32_t)(f + 2 - data);
            q = f + 2;
            continue;
        }
        end = f;                                                       // EOI or another marker: the entropy-coded data ends here
        break;
    }
    if (found != n_int - 1) return -3;
    h[27] = (int32_t)(end - data);
    return need;
}

}  // extern "C"
