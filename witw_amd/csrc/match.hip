// Fused orientation search + chord distance (gfx950, fp32 MFMA).
//
// Replaces the reference's correlation -> crop_overhead -> l2_distance chain
// (model/cvig_fov.py:297-363, called at :450-453 and :547-549) WITHOUT materialising the
// [Bo,Bs,C,H,We] crop tensor (268 MB at B=128, 17 GB at B=1024):
//   score[o,s,shift] = sum_{ch,k<We} ov[o,ch,(k+shift)%64] * su[s,ch,k]        (ch = c*4+h, 64 rows)
//   orientation[o,s] = argmax_shift score   (first index on ties, as torch.argmax)
//   distance[o,s]    = 2*(1 - score_max / (|window(o,orientation)| * |su[s]|))
// As a GEMM per overhead image: M = surfaces, N = 64 shifts, K = 64*We. The surface tile is the
// A operand (LDS rows of stride 65 floats -> conflict-free ds_read_b32), the overhead row,
// stored twice back to back so that (k+shift) never wraps, is the B operand read at consecutive
// addresses by the 32 lanes of a shift tile. Accumulators hold [surface rows][shift lanes]; the
// arg-max over shifts is a 5-step wavefront butterfly per accumulator register.
#include "common.h"

namespace {

constexpr int MS = 128;        // surfaces per block (4 M-tiles)
constexpr int SUS = 65;        // LDS row stride of the surface tile (floats)
constexpr int NT = 256;
#ifndef NSPLIT_RPC
#define NSPLIT_RPC 1           // match_kernel_nsplit: embedding rows per LDS stage (1: 33 KB of LDS; 2: 68 KB and half the barriers -- measured 94 against 92 us at 128 x 128)
#endif

struct MatchArgs {
    const float* ov;     // [Bo,64,64]
    const float* su;     // [Bs,64,We]
    const float* wn;     // [Bo,64] window norms per shift
    const float* sn;     // [Bs]    surface norms
    long long* orientation;  // [Bo,Bs] or null
    float* distance;         // [Bo,Bs] or null
    float* score;            // [Bo,Bs] or null (max correlation)
    int Bo, Bs, We;
};

// OPW = overhead images per wave (1 or 2), MT = 32-surface M-tiles per wave (2, or 1 for batches too small to fill the
// chip with 128-surface blocks); a block covers 2*OPW overheads x 64*MT surfaces.
template <int OPW, int MT>
__global__ __launch_bounds__(NT) void match_kernel(MatchArgs p) {
    constexpr int MO = 2 * OPW;
    constexpr int MSB = 64 * MT;         // surfaces per block
    constexpr int SU_F = MSB * SUS;      // floats per surface stage
    constexpr int RPW = MSB / 4;         // surface rows staged per wave
    constexpr int OV_F = MO * 128;       // floats per overhead stage
    __shared__ float smem[2 * (SU_F + OV_F)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hk = lane >> 5;
    const int s0 = blockIdx.x * MSB;
    const int o0 = blockIdx.y * MO;
    const int We = p.We;
    const int Wp = (We + 1) & ~1;        // K per row padded to the MFMA k-step of 2
    const int wm = wave >> 1;            // surface half: rows [32*MT*wm, 32*MT*(wm+1))
    const int wo = wave & 1;             // overhead group: local overheads [OPW*wo, OPW*wo+OPW)

    // staging roles: lane <-> k within a row, (wave + 4*i) <-> surface row
    const bool kin = lane < We;
    float rsu[RPW];
    float rov = 0.f;
    const int ovo = tid >> 6, ovw = tid & 63;   // overhead staging: thread -> (local overhead, column)
    auto load_stage = [&](int r) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int s = s0 + wave + 4 * i;
            float v = 0.f;
            if (kin && s < p.Bs) v = p.su[((size_t)s * 64 + r) * We + lane];
            rsu[i] = v;
        }
        rov = 0.f;
        if (ovo < MO && o0 + ovo < p.Bo) rov = p.ov[((size_t)(o0 + ovo) * 64 + r) * 64 + ovw];
    };
    auto store_stage = [&](int buf) {
        float* su_s = smem + buf * (SU_F + OV_F);
        float* ov_s = su_s + SU_F;
        if (lane < Wp) {
#pragma unroll
            for (int i = 0; i < RPW; ++i) su_s[(wave + 4 * i) * SUS + lane] = rsu[i];
        }
        if (ovo < MO) {
            ov_s[ovo * 128 + ovw] = rov;
            ov_s[ovo * 128 + 64 + ovw] = rov;
        }
    };

    f32x16 acc[MT][OPW][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < OPW; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int arow0 = (32 * MT * wm + l31) * SUS + hk;
    const int arow1 = arow0 + 32 * SUS;
    const int bcol = OPW * wo * 128 + l31 + hk;
    for (int r = 0; r < 64; ++r) {
        const int cur = r & 1;
        if (r + 1 < 64) load_stage(r + 1);
        const float* su_s = smem + cur * (SU_F + OV_F);
        const float* ov_s = su_s + SU_F;
#pragma unroll 4
        for (int k = 0; k < Wp; k += 2) {
            const float a0 = su_s[arow0 + k];
            const float a1 = (MT == 2) ? su_s[arow1 + k] : 0.f;
            float b[OPW][2];
#pragma unroll
            for (int o = 0; o < OPW; ++o) {
                b[o][0] = ov_s[bcol + o * 128 + k];
                b[o][1] = ov_s[bcol + o * 128 + k + 32];
            }
#pragma unroll
            for (int o = 0; o < OPW; ++o)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[0][o][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[o][n], acc[0][o][n], 0, 0, 0);
                    if (MT == 2) acc[MT - 1][o][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[o][n], acc[MT - 1][o][n], 0, 0, 0);
                }
        }
        if (r + 1 < 64) store_stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: arg-max over the 64 shifts (2 N-tiles x 32 lanes), first index wins ties
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int o = 0; o < OPW; ++o) {
            const int og = o0 + OPW * wo + o;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][o][0][r];
                int idx = l31;
                const float v1 = acc[mt][o][1][r];
                if (v1 > v) { v = v1; idx = 32 + l31; }
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const float vo = __shfl_xor(v, d, 64);
                    const int io = __shfl_xor(idx, d, 64);
                    if (vo > v || (vo == v && io < idx)) { v = vo; idx = io; }
                }
                const int srow = s0 + 32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hk;
                if (l31 == r && og < p.Bo && srow < p.Bs) {
                    const size_t off = (size_t)og * p.Bs + srow;
                    if (p.orientation) p.orientation[off] = idx;
                    if (p.score) p.score[off] = v;
                    if (p.distance) p.distance[off] = 2.f * (1.f - v / (p.wn[(size_t)og * 64 + idx] * p.sn[srow]));
                }
            }
        }
}


// ---- a single minibatch (128 x 128 pairs: the matching step of BASELINE configs[1] / configs[3]): match_kernel<1,1> makes 128
// workgroups of 4 waves there, half a wave per SIMD of the chip. Here a workgroup is ONE overhead x 64 surfaces and its four waves
// split surfaces (2 halves of 32) x SHIFTS (2 halves of 32): 256 workgroups, one wave on every SIMD, one accumulator tile per
// wave. The two shift halves of a (surface, overhead) pair meet in the epilogue through LDS: larger score wins, the lower
// half (smaller shift) on a tie -- the first-index rule of torch.argmax. Every score is the same k-ordered fma chain as in
// match_kernel (K is not split), so orientation / score / distance carry the same bits.
template <int WP, int RPC>      // WP: surface columns per row, zero-padded: 16, 32 or 64 (every k-loop fully unrolled); RPC: rows per LDS stage
__global__ __launch_bounds__(NT) void match_kernel_nsplit(MatchArgs p) {
    constexpr int MSB = 64;              // surfaces per block
    constexpr int SUC = RPC * 64 + 1;    // LDS stride of a surface's RPC rows (odd: conflict-free ds_read_b32 down the surfaces)
    constexpr int SU_F = MSB * SUC;
    constexpr int RPW = MSB / 4;
    constexpr int OV_F = RPC * 128;
    extern __shared__ float smem[];      // 2 stages of SU_F + OV_F floats (68 KB at RPC = 2)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hk = lane >> 5;
    const int s0 = blockIdx.x * MSB;
    const int og = blockIdx.y;
    const int We = p.We;
    constexpr int Wp = WP;               // zero columns beyond We add exact zeros to every score: the same bits as K = We
    const int wm = wave >> 1;            // surface half: rows [32*wm, 32*wm+32)
    const int wn = wave & 1;             // shift half: shifts [32*wn, 32*wn+32)

    // staging: the rows of the two embeddings travel global -> registers -> LDS, TWO chunks (of RPC rows) ahead of the MFMAs (two
    // register sets A / B): one row's MFMAs last 0.85 us at one wave per SIMD, less than a global load takes to come back. Buffer
    // loads: the row offset is a scalar, lanes beyond We and surfaces beyond Bs carry an out-of-range offset and read 0 -- no
    // VALU work or branch per load.
    constexpr unsigned OOR = 0x80000000u;
    const int rows_here = min(MSB, p.Bs - s0);
    __amdgpu_buffer_rsrc_t su_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.su + (size_t)s0 * 64 * We), 0,
                                                                     (unsigned)rows_here * 64u * We * 4u, 0x00020000);
    __amdgpu_buffer_rsrc_t ov_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ov + (size_t)og * 4096), 0, 4096u * 4u, 0x00020000);
    unsigned suoff[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = wave + 4 * i;
        suoff[i] = (lane < We && row < rows_here) ? ((unsigned)row * 64u * We + lane) * 4u : OOR;
    }
    const unsigned ovoff = (unsigned)lane * 4u;
    constexpr int NCH = 64 / RPC;        // chunks
    float rsuA[RPC][RPW], rsuB[RPC][RPW];
    float rovA[RPC], rovB[RPC];
    auto load_A = [&](int c) {
#pragma unroll
        for (int rr = 0; rr < RPC; ++rr) {
            const unsigned srow = (unsigned)(c * RPC + rr) * We * 4u;
#pragma unroll
            for (int i = 0; i < RPW; ++i) rsuA[rr][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(su_rs, suoff[i], srow, 0));
            rovA[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ov_rs, ovoff, (unsigned)(c * RPC + rr) * 256u, 0));
        }
    };
    auto load_B = [&](int c) {
#pragma unroll
        for (int rr = 0; rr < RPC; ++rr) {
            const unsigned srow = (unsigned)(c * RPC + rr) * We * 4u;
#pragma unroll
            for (int i = 0; i < RPW; ++i) rsuB[rr][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(su_rs, suoff[i], srow, 0));
            rovB[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ov_rs, ovoff, (unsigned)(c * RPC + rr) * 256u, 0));
        }
    };
    auto store_A = [&](int buf) {
        float* su_s = smem + buf * (SU_F + OV_F);
        float* ov_s = su_s + SU_F;
#pragma unroll
        for (int rr = 0; rr < RPC; ++rr) {
            if (lane < Wp) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) su_s[(wave + 4 * i) * SUC + rr * 64 + lane] = rsuA[rr][i];
            }
            if (wave == 0) { ov_s[rr * 128 + lane] = rovA[rr]; ov_s[rr * 128 + 64 + lane] = rovA[rr]; }
        }
    };
    auto store_B = [&](int buf) {
        float* su_s = smem + buf * (SU_F + OV_F);
        float* ov_s = su_s + SU_F;
#pragma unroll
        for (int rr = 0; rr < RPC; ++rr) {
            if (lane < Wp) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) su_s[(wave + 4 * i) * SUC + rr * 64 + lane] = rsuB[rr][i];
            }
            if (wave == 0) { ov_s[rr * 128 + lane] = rovB[rr]; ov_s[rr * 128 + 64 + lane] = rovB[rr]; }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    load_A(0);
    load_B(1);
    store_A(0);
    __syncthreads();

    // a chunk's k-steps (row after row: the k-order of the other kernels) in four groups; the operands of group g+1 are read (one
    // LDS read pair per MFMA) while group g multiplies, the chunk's barrier sits in front of the last group and the first group of
    // the NEXT chunk is read behind it
    constexpr int KS = Wp / 2, KSC = RPC * KS, GS = KSC / 4;
    const int arow = (32 * wm + l31) * SUC + hk;
    const int bcol = 32 * wn + l31 + hk;
    float fa[2][GS], fb[2][GS];
    auto read_group = [&](int set, int stage, int g) {
        const float* su_s = smem + stage * (SU_F + OV_F);
        const float* ov_s = su_s + SU_F;
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int kk = GS * g + j, rr = kk / KS, k = kk - rr * KS;
            fa[set][j] = su_s[arow + rr * 64 + 2 * k];
            fb[set][j] = ov_s[bcol + rr * 128 + 2 * k];
        }
    };
    auto mfma_group = [&](int set) {
#pragma unroll
        for (int j = 0; j < GS; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][j], fb[set][j], acc, 0, 0, 0);
    };
    // issue order inside a group: GS x (one MFMA, its two LDS reads for the next group), the group's global loads (mask 0x020)
    // or LDS writes (0x200) spread evenly behind the MFMAs
    constexpr int PER = (RPC * (RPW + 1) + GS - 1) / GS;
#define NSPLIT_INTERLEAVE(OTHER_MASK)                                                   \
    _Pragma("unroll") for (int j = 0; j < GS; ++j) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                               \
        if (OTHER_MASK) __builtin_amdgcn_sched_group_barrier(OTHER_MASK, PER, 0);        \
    }
    read_group(0, 0, 0);
    for (int c = 0; c < NCH; c += 2) {
        // chunk c from stage 0; set B holds chunk c+1 (loaded one chunk ago), set A takes chunk c+2
        load_A(min(c + 2, NCH - 1));
        read_group(1, 0, 1); mfma_group(0); NSPLIT_INTERLEAVE(0x020)
        read_group(0, 0, 2); store_B(1); mfma_group(1); NSPLIT_INTERLEAVE(0x200)
        read_group(1, 0, 3); mfma_group(0); NSPLIT_INTERLEAVE(0)
        __syncthreads();
        read_group(0, 1, 0); mfma_group(1); NSPLIT_INTERLEAVE(0)
        // chunk c+1 from stage 1; set A holds chunk c+2, set B takes chunk c+3
        load_B(min(c + 3, NCH - 1));
        read_group(1, 1, 1); mfma_group(0); NSPLIT_INTERLEAVE(0x020)
        read_group(0, 1, 2); store_A(0); mfma_group(1); NSPLIT_INTERLEAVE(0x200)
        read_group(1, 1, 3); mfma_group(0); NSPLIT_INTERLEAVE(0)
        __syncthreads();
        read_group(0, 0, 0); mfma_group(1); NSPLIT_INTERLEAVE(0)
    }

#undef NSPLIT_INTERLEAVE
    // ---- epilogue: arg-max over this wave's 32 shifts, then across the two shift halves (LDS), first index wins ties
    float* xv = smem;                                          // [2 surface halves][32 rows] best score of the upper shift half
    int* xi = reinterpret_cast<int*>(smem + 64);               // ... and its shift
    float bv[16];
    int bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        int idx = 32 * wn + l31;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const float vo = __shfl_xor(v, d, 64);
            const int io = __shfl_xor(idx, d, 64);
            if (vo > v || (vo == v && io < idx)) { v = vo; idx = io; }
        }
        bv[r] = v;
        bi[r] = idx;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hk;      // surface row inside this wave's half
        if (wn == 1 && l31 == r) {
            xv[32 * wm + row] = v;
            xi[32 * wm + row] = idx;
        }
    }
    __syncthreads();
    if (wn == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hk;
            const int srow = s0 + 32 * wm + row;
            if (l31 == r && srow < p.Bs) {
                float v = bv[r];
                int idx = bi[r];
                const float v1 = xv[32 * wm + row];
                if (v1 > v) { v = v1; idx = xi[32 * wm + row]; }
                const size_t off = (size_t)og * p.Bs + srow;
                if (p.orientation) p.orientation[off] = idx;
                if (p.score) p.score[off] = v;
                if (p.distance) p.distance[off] = 2.f * (1.f - v / (p.wn[(size_t)og * 64 + idx] * p.sn[srow]));
            }
        }
    }
}


// ---- pipelined variant for full-width surfaces (We in {63,64}: 32 MFMA k-steps per embedding row), the
// retrieval shape (BASELINE config C5). Same tiling as match_kernel<2> (4 overheads x 128 surfaces per
// workgroup) but: staging uses buffer loads (row offset in a scalar, out-of-range rows/columns read 0: no
// per-load VALU next to the f32 MFMAs), the 32 k-steps are fully unrolled with the operands of step k+1
// read during step k, every LDS / global instruction is issued alone between two MFMAs
// (sched_group_barrier), and the per-row barrier sits in front of the last k-step so the next row's first
// operands arrive behind it.
typedef unsigned int u32x1;

__global__ __launch_bounds__(NT, 2) void match_kernel_w64(MatchArgs p) {
    constexpr int OPW = 2, MO = 4, KS = 32;
    constexpr int SU_F = MS * SUS;
    constexpr int OV_F = MO * 128;
    constexpr unsigned OOR = 0x80000000u;
    __shared__ float smem[2 * (SU_F + OV_F)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hk = lane >> 5;
    const int s0 = blockIdx.x * MS;
    const int o0 = blockIdx.y * MO;
    const int We = p.We;
    const int wm = wave >> 1, wo = wave & 1;

    // staging: lane <-> k within a row, (wave + 4*i) <-> surface row; offsets relative to the tile's first row
    const int rows_here = min(MS, p.Bs - s0);
    const unsigned su_bytes = (unsigned)rows_here * 64u * We * 4u;
    __amdgpu_buffer_rsrc_t su_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.su + (size_t)s0 * 64 * We), 0, su_bytes, 0x00020000);
    const int ov_here = min(MO, p.Bo - o0);
    __amdgpu_buffer_rsrc_t ov_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.ov + (size_t)o0 * 4096), 0, (unsigned)ov_here * 4096u * 4u, 0x00020000);
    unsigned suoff[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int row = wave + 4 * i;
        suoff[i] = (lane < We && row < rows_here) ? ((unsigned)row * 64u * We + lane) * 4u : OOR;
    }
    const int ovo = tid >> 6, ovw = tid & 63;
    const unsigned ovoff = (ovo < ov_here) ? ((unsigned)ovo * 4096u + ovw) * 4u : OOR;

    float rsu[32];
    float rov;
    auto load_stage = [&](int r) {
        const unsigned srow = (unsigned)r * We * 4u;
#pragma unroll
        for (int i = 0; i < 32; ++i) rsu[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(su_rs, suoff[i], srow, 0));
        rov = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ov_rs, ovoff, (unsigned)r * 256u, 0));
    };
    auto store_stage = [&](int buf) {
        float* su_s = smem + buf * (SU_F + OV_F);
        float* ov_s = su_s + SU_F;
#pragma unroll
        for (int i = 0; i < 32; ++i) su_s[(wave + 4 * i) * SUS + lane] = rsu[i];
        ov_s[ovo * 128 + ovw] = rov;
        ov_s[ovo * 128 + 64 + ovw] = rov;
    };

    f32x16 acc[2][OPW][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < OPW; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

    const int arow0 = (64 * wm + l31) * SUS + hk;
    const int arow1 = arow0 + 32 * SUS;
    const int bcol = OPW * wo * 128 + l31 + hk;
    float fa[2][2], fb[2][OPW][2];
    auto read_frags = [&](int set, const float* su_s, const float* ov_s, int k) {
        fa[set][0] = su_s[arow0 + 2 * k];
        fa[set][1] = su_s[arow1 + 2 * k];
#pragma unroll
        for (int o = 0; o < OPW; ++o) {
            fb[set][o][0] = ov_s[bcol + o * 128 + 2 * k];
            fb[set][o][1] = ov_s[bcol + o * 128 + 2 * k + 32];
        }
    };
    auto mfma_step = [&](int set) {
#pragma unroll
        for (int o = 0; o < OPW; ++o)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[0][o][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][0], fb[set][o][n], acc[0][o][n], 0, 0, 0);
                acc[1][o][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][1], fb[set][o][n], acc[1][o][n], 0, 0, 0);
            }
    };

    load_stage(0);
    store_stage(0);
    __syncthreads();
    read_frags(0, smem, smem + SU_F, 0);

    for (int r = 0; r < 64; ++r) {
        const int cur = r & 1;
        const int rn = (r + 1 < 64) ? r + 1 : r;      // the last row restages itself (never read)
        const float* su_s = smem + cur * (SU_F + OV_F);
        const float* ov_s = su_s + SU_F;
        const float* su_n = smem + (cur ^ 1) * (SU_F + OV_F);
        load_stage(rn);
#pragma unroll
        for (int k = 0; k < KS - 1; ++k) {
            read_frags((k + 1) & 1, su_s, ov_s, k + 1);
            if (k == 17) store_stage(cur ^ 1);
            mfma_step(k & 1);
            // 8 MFMAs per k-step: 6 operand reads + (k < 17: 2 global loads | k == 17: the 34 LDS writes)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (k < 17) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        __syncthreads();
        read_frags(0, su_n, su_n + SU_F, 0);
        mfma_step(1);                                 // k-step 31 (set (KS-1)&1 == 1)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }

    // ---- epilogue: identical to match_kernel
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int o = 0; o < OPW; ++o) {
            const int og = o0 + OPW * wo + o;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][o][0][r];
                int idx = l31;
                const float v1 = acc[mt][o][1][r];
                if (v1 > v) { v = v1; idx = 32 + l31; }
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const float vo = __shfl_xor(v, d, 64);
                    const int io = __shfl_xor(idx, d, 64);
                    if (vo > v || (vo == v && io < idx)) { v = vo; idx = io; }
                }
                const int srow = s0 + 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hk;
                if (l31 == r && og < p.Bo && srow < p.Bs) {
                    const size_t off = (size_t)og * p.Bs + srow;
                    if (p.orientation) p.orientation[off] = idx;
                    if (p.score) p.score[off] = v;
                    if (p.distance) p.distance[off] = 2.f * (1.f - v / (p.wn[(size_t)og * 64 + idx] * p.sn[srow]));
                }
            }
        }
}

// ---- narrower surfaces (We < 63: fov < 355): the same pipelined loop as match_kernel_w64, but one LDS stage holds
// R = 4 (We <= 16), 2 (We <= 32) or 1 (We <= 62) embedding rows side by side, each zero-padded to WP = 64/R columns, so that a
// stage is again 32 full k-steps (a stage per row would be 6 k-steps and a barrier at We = 12). K index of a
// stage = rr*WP + k; the B operand of (rr, k) is overhead row rr at column k + shift, overhead rows are stored
// 64 + WP floats long (the window never wraps). Zero columns add exact zeros: scores are bit-identical to the
// other kernels.
template <int R>
__global__ __launch_bounds__(NT, 2) void match_kernel_rows(MatchArgs p) {
    constexpr int OPW = 2, MO = 4, KS = 32, WP = 64 / R, OVS = 64 + WP;
    constexpr int SU_F = MS * SUS;
    constexpr int OV_F = MO * R * OVS;
    constexpr unsigned OOR = 0x80000000u;
    __shared__ float smem[2 * (SU_F + OV_F)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hk = lane >> 5;
    const int s0 = blockIdx.x * MS;
    const int o0 = blockIdx.y * MO;
    const int We = p.We;
    const int wm = wave >> 1, wo = wave & 1;

    // staging: lane <-> (row rr = lane / WP of the stage, column k = lane % WP), (wave + 4*i) <-> surface
    const int rows_here = min(MS, p.Bs - s0);
    __amdgpu_buffer_rsrc_t su_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.su + (size_t)s0 * 64 * We), 0,
                                                                    (unsigned)rows_here * 64u * We * 4u, 0x00020000);
    const int ov_here = min(MO, p.Bo - o0);
    __amdgpu_buffer_rsrc_t ov_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.ov + (size_t)o0 * 4096), 0, (unsigned)ov_here * 4096u * 4u, 0x00020000);
    const int lrr = lane / WP, lk = lane % WP;
    unsigned suoff[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int row = wave + 4 * i;
        suoff[i] = (lk < We && row < rows_here) ? ((unsigned)row * 64u * We + lrr * We + lk) * 4u : OOR;
    }
    // overhead staging: R passes, thread -> (local overhead tid/64, column tid%64), pass <-> row rr
    const int ovo = tid >> 6, ovw = tid & 63;
    const unsigned ovoff = (ovo < ov_here) ? ((unsigned)ovo * 4096u + ovw) * 4u : OOR;

    float rsu[32];
    float rov[R];
    auto load_stage = [&](int st) {
        const unsigned srow = (unsigned)st * R * We * 4u;
#pragma unroll
        for (int i = 0; i < 32; ++i) rsu[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(su_rs, suoff[i], srow, 0));
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
            rov[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ov_rs, ovoff, (unsigned)(st * R + rr) * 256u, 0));
    };
    auto store_stage = [&](int buf) {
        float* su_s = smem + buf * (SU_F + OV_F);
        float* ov_s = su_s + SU_F;
#pragma unroll
        for (int i = 0; i < 32; ++i) su_s[(wave + 4 * i) * SUS + lane] = rsu[i];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            ov_s[(ovo * R + rr) * OVS + ovw] = rov[rr];
            if (ovw < WP) ov_s[(ovo * R + rr) * OVS + 64 + ovw] = rov[rr];
        }
    };

    f32x16 acc[2][OPW][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < OPW; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

    const int arow0 = (64 * wm + l31) * SUS + hk;
    const int arow1 = arow0 + 32 * SUS;
    const int bcol = OPW * wo * R * OVS + l31 + hk;
    float fa[2][2], fb[2][OPW][2];
    auto read_frags = [&](int set, const float* su_s, const float* ov_s, int k) {
        const int rr = (2 * k) / WP, kk = (2 * k) % WP;       // compile-time after unrolling
        fa[set][0] = su_s[arow0 + 2 * k];
        fa[set][1] = su_s[arow1 + 2 * k];
#pragma unroll
        for (int o = 0; o < OPW; ++o) {
            fb[set][o][0] = ov_s[bcol + (o * R + rr) * OVS + kk];
            fb[set][o][1] = ov_s[bcol + (o * R + rr) * OVS + kk + 32];
        }
    };
    auto mfma_step = [&](int set) {
#pragma unroll
        for (int o = 0; o < OPW; ++o)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[0][o][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][0], fb[set][o][n], acc[0][o][n], 0, 0, 0);
                acc[1][o][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][1], fb[set][o][n], acc[1][o][n], 0, 0, 0);
            }
    };

    constexpr int NS = 64 / R;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    read_frags(0, smem, smem + SU_F, 0);

    for (int st = 0; st < NS; ++st) {
        const int cur = st & 1;
        const int sn = (st + 1 < NS) ? st + 1 : st;      // the last stage restages itself (never read)
        const float* su_s = smem + cur * (SU_F + OV_F);
        const float* ov_s = su_s + SU_F;
        const float* su_n = smem + (cur ^ 1) * (SU_F + OV_F);
        load_stage(sn);
#pragma unroll
        for (int k = 0; k < KS - 1; ++k) {
            read_frags((k + 1) & 1, su_s, ov_s, k + 1);
            if (k == 17) store_stage(cur ^ 1);
            mfma_step(k & 1);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (k < 17) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        __syncthreads();
        read_frags(0, su_n, su_n + SU_F, 0);
        mfma_step(1);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }

    // ---- epilogue: identical to match_kernel
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int o = 0; o < OPW; ++o) {
            const int og = o0 + OPW * wo + o;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][o][0][r];
                int idx = l31;
                const float v1 = acc[mt][o][1][r];
                if (v1 > v) { v = v1; idx = 32 + l31; }
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const float vo = __shfl_xor(v, d, 64);
                    const int io = __shfl_xor(idx, d, 64);
                    if (vo > v || (vo == v && io < idx)) { v = vo; idx = io; }
                }
                const int srow = s0 + 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hk;
                if (l31 == r && og < p.Bo && srow < p.Bs) {
                    const size_t off = (size_t)og * p.Bs + srow;
                    if (p.orientation) p.orientation[off] = idx;
                    if (p.score) p.score[off] = v;
                    if (p.distance) p.distance[off] = 2.f * (1.f - v / (p.wn[(size_t)og * 64 + idx] * p.sn[srow]));
                }
            }
        }
}

// wn[o][shift] = sqrt(sum_{ch} sum_{k<We} ov[o][ch][(k+shift)%64]^2): the L2 norm of the window
// that crop_overhead would cut at that shift (model/cvig_fov.py:335-341,350-351).
__global__ __launch_bounds__(256) void window_norm_kernel(const float* __restrict__ ov, float* __restrict__ wn, int We) {
    __shared__ float part[4][64];
    __shared__ float col[64];
    const int o = blockIdx.x, t = threadIdx.x, w = t & 63, g = t >> 6;
    const float* base = ov + (size_t)o * 4096;
    float s = 0.f;
    for (int ch = g * 16; ch < g * 16 + 16; ++ch) {
        const float v = base[ch * 64 + w];
        s += v * v;
    }
    part[g][w] = s;
    __syncthreads();
    if (t < 64) col[t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    __syncthreads();
    if (t < 64) {
        float acc = 0.f;
        for (int k = 0; k < We; ++k) acc += col[(t + k) & 63];
        wn[(size_t)o * 64 + t] = sqrtf(acc);
    }
}

// sn[s] = |su[s]|_2 over all 64*We elements (model/cvig_fov.py:356-357).
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
    __shared__ float part[4];
    const float* base = x + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = base[i];
        s += v * v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sqrtf((part[0] + part[1]) + (part[2] + part[3]));
}

// crop_overhead as a materialising gather (model/cvig_fov.py:318-343); compatibility entry for
// callers that index the crop (tools/heatmap/heatmap.py:173, TensorBoard dumps). One thread per
// output element, k fastest.
__global__ void crop_overhead_kernel(const float* __restrict__ ov, const long long* __restrict__ ori,
                                     float* __restrict__ out, int Bo, int Bs, int We, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = idx % We;
    size_t t = idx / We;
    const int ch = t % 64;
    t /= 64;
    const int s = t % Bs;
    const int o = (int)(t / Bs);
    const int sh = (int)ori[(size_t)o * Bs + s];
    out[idx] = ov[((size_t)o * 64 + ch) * 64 + ((k + sh) & 63)];
}

// l2_distance on a materialised crop (model/cvig_fov.py:346-363): one block per (o,s).
__global__ __launch_bounds__(256) void l2_distance_kernel(const float* __restrict__ crop, const float* __restrict__ su,
                                                           float* __restrict__ dist, int Bs, int n) {
    __shared__ float part[3][4];
    const int s = blockIdx.x % Bs;
    const float* a = crop + (size_t)blockIdx.x * n;
    const float* b = su + (size_t)s * n;
    float aa = 0.f, bb = 0.f, ab = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = a[i], y = b[i];
        aa += x * x;
        bb += y * y;
        ab += x * y;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        aa += __shfl_xor(aa, d, 64);
        bb += __shfl_xor(bb, d, 64);
        ab += __shfl_xor(ab, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = aa;
        part[1][threadIdx.x >> 6] = bb;
        part[2][threadIdx.x >> 6] = ab;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float na = sqrtf((part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
        const float nb = sqrtf((part[1][0] + part[1][1]) + (part[1][2] + part[1][3]));
        const float dot = (part[2][0] + part[2][1]) + (part[2][2] + part[2][3]);
        dist[blockIdx.x] = 2.f * (1.f - dot / (na * nb));
    }
}

// rank[q] = #{o : D[o][q] <= D[true(q)][q]}  (model/cvig_fov.py:550-552); true(q) = q + true_offset.
// grid (ceil(Bs/256), chunks of gallery rows); integer atomics -> order independent, exact.
__global__ __launch_bounds__(256) void rank_count_kernel(const float* __restrict__ D, int* __restrict__ ranks, int Bo, int Bs,
                                                         int true_offset, int rows_per_block) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Bs) return;
    const int t = q + true_offset;
    if (t < 0 || t >= Bo) return;
    const float dt = D[(size_t)t * Bs + q];
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(Bo, r0 + rows_per_block);
    int c = 0;
    for (int o = r0; o < r1; ++o) c += (D[(size_t)o * Bs + q] <= dt) ? 1 : 0;
    if (c) atomicAdd(&ranks[q], c);
}


// ---- backward of the fused match (autograd of correlation->crop->l2_distance, reference
// model/cvig_fov.py:450-453; the arg-max orientation is a constant of the graph, as in torch):
//   d = 2*(1 - c/(wn*sn)),  c = <window, su>,  wn = |window|, sn = |su|
//   dd/dsu[k]     = -2*( window[k]/(wn*sn) - c*su[k]/(wn*sn^3) )
//   dd/dwindow[k] = -2*( su[k]/(wn*sn)     - c*window[k]/(wn^3*sn) )
// grad_su: one block per surface s, loops over overheads; grad_ov: one block per overhead o,
// loops over surfaces and scatters through the rotation. fp32 VALU (Bo*Bs*E MACs, 1/64 of forward).
__global__ __launch_bounds__(256) void match_bwd_su_kernel(const float* __restrict__ ov, const float* __restrict__ su,
                                                            const long long* __restrict__ ori, const float* __restrict__ score,
                                                            const float* __restrict__ wn, const float* __restrict__ sn,
                                                            const float* __restrict__ gD, float* __restrict__ gsu, int Bo,
                                                            int Bs, int We, int o_per_split, float* __restrict__ scratch) {
    // grid (Bs, splits): split y sums the overheads [y*o_per_split, (y+1)*o_per_split). With one split the result goes
    // straight to gsu; otherwise partial sums [y][s][E] and the partial self terms [splits][Bs] behind them go to
    // scratch and match_bwd_su_finish_kernel adds them in a fixed order (a small batch has too few surfaces to fill the chip).
    __shared__ float coef[256];
    __shared__ int rot[256];
    __shared__ float part[4];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int E = 64 * We;
    const int o_begin = blockIdx.y * o_per_split, o_end = min(Bo, o_begin + o_per_split);
    const float sns = sn[s];
    int ch[16], kk[16];
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;
        ch[i] = e / We;
        kk[i] = e - ch[i] * We;
        acc[i] = 0.f;
    }
    float self = 0.f;   // sum_o gD*2c/(wn*sn^3), accumulated by thread 0..255 over its o's then reduced
    for (int o0 = o_begin; o0 < o_end; o0 += 256) {
        const int o = o0 + tid;
        __syncthreads();
        if (o < o_end) {
            const size_t off = (size_t)o * Bs + s;
            const int t = (int)ori[off];
            const float w = wn[(size_t)o * 64 + t];
            const float g = gD[off];
            coef[tid] = g * (-2.f / (w * sns));
            rot[tid] = t;
            self += g * (2.f * score[off] / (w * sns * sns * sns));
        }
        __syncthreads();
        const int n = min(256, o_end - o0);
        for (int j = 0; j < n; ++j) {
            const float a = coef[j];
            const int t = rot[j];
            const float* row = ov + (size_t)(o0 + j) * 4096;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (tid + 256 * i < E) acc[i] += a * row[ch[i] * 64 + ((kk[i] + t) & 63)];
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) self += __shfl_xor(self, d, 64);
    __syncthreads();
    if ((tid & 63) == 0) part[tid >> 6] = self;
    __syncthreads();
    const float selfsum = (part[0] + part[1]) + (part[2] + part[3]);
    if (gridDim.y == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            if (e < E) gsu[(size_t)s * E + e] = acc[i] + su[(size_t)s * E + e] * selfsum;
        }
    } else {
        float* dst = scratch + ((size_t)blockIdx.y * Bs + s) * E;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            if (e < E) dst[e] = acc[i];
        }
        if (tid == 0) scratch[(size_t)gridDim.y * Bs * E + (size_t)blockIdx.y * Bs + s] = selfsum;
    }
}

__global__ __launch_bounds__(256) void match_bwd_su_finish_kernel(const float* __restrict__ su, const float* __restrict__ scratch,
                                                                   float* __restrict__ gsu, int Bs, int E, int splits) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)Bs * E) return;
    const int s = (int)(idx / E);
    float a = 0.f, self = 0.f;
    for (int y = 0; y < splits; ++y) {
        a += scratch[(size_t)y * Bs * E + idx];
        self += scratch[(size_t)splits * Bs * E + (size_t)y * Bs + s];
    }
    gsu[idx] = a + su[idx] * self;
}

__global__ __launch_bounds__(256) void match_bwd_ov_kernel(const float* __restrict__ ov, const float* __restrict__ su,
                                                            const long long* __restrict__ ori, const float* __restrict__ score,
                                                            const float* __restrict__ wn, const float* __restrict__ sn,
                                                            const float* __restrict__ gD, float* __restrict__ gov, int Bo,
                                                            int Bs, int We) {
    __shared__ float coef[256];
    __shared__ float coef2[256];
    __shared__ int rot[256];
    const int o = blockIdx.x, tid = threadIdx.x;
    const int w = tid & 63, cg = tid >> 6;   // element (ch = cg + 4*i, w)
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float beta = 0.f;                        // sum_s gD*2c/(wn^3*sn) over surfaces whose window covers column w
    for (int s0 = 0; s0 < Bs; s0 += 256) {
        const int s = s0 + tid;
        __syncthreads();
        if (s < Bs) {
            const size_t off = (size_t)o * Bs + s;
            const int t = (int)ori[off];
            const float wv = wn[(size_t)o * 64 + t];
            const float g = gD[off];
            const float sv = sn[s];
            coef[tid] = g * (-2.f / (wv * sv));
            coef2[tid] = g * (2.f * score[off] / (wv * wv * wv * sv));
            rot[tid] = t;
        }
        __syncthreads();
        const int n = min(256, Bs - s0);
        for (int j = 0; j < n; ++j) {
            const int k = (w - rot[j]) & 63;
            if (k < We) {
                const float a = coef[j];
                beta += coef2[j];
                const float* base = su + (size_t)(s0 + j) * 64 * We + k;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += a * base[(cg + 4 * i) * We];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const size_t e = (size_t)o * 4096 + (cg + 4 * i) * 64 + w;
        gov[e] = acc[i] + ov[e] * beta;
    }
}

// zero fill as a kernel (not hipMemsetAsync): a launch is captured faithfully when the step is recorded into a hipGraph
// (parallel.CapturedStep); a memset node of this size replayed with stale contents on ROCm 7.2
__global__ void zero_i32_kernel(int* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}

// Sharded-gallery form: count against an explicit per-query threshold (the true match's distance,
// produced by whichever rank owns that gallery row).
__global__ __launch_bounds__(256) void rank_count_thresh_kernel(const float* __restrict__ D, const float* __restrict__ thr,
                                                                int* __restrict__ ranks, int Bo, int Bs, int rows_per_block) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Bs) return;
    const float dt = thr[q];
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(Bo, r0 + rows_per_block);
    int c = 0;
    for (int o = r0; o < r1; ++o) c += (D[(size_t)o * Bs + q] <= dt) ? 1 : 0;
    if (c) atomicAdd(&ranks[q], c);
}

// ---- a LIST of (overhead, surface) pairs through exactly the arithmetic of the all-pairs kernels above: per pair the same
// chain of v_mfma_f32_32x32x2_f32 over (row, k pair) from a zero accumulator, the same first-index arg-max, the same distance
// expression -- bit-identical to what witw_match_fwd writes for that pair. One wave per pair: all 32 A rows carry the pair's
// surface (31/32 of the matrix work is redundant: the price of bit-identity; the callers re-score a few pairs per million).
// Used by the index-exact spectral retrieval (cvig_fov.retrieve(method='dft')) for the pairs whose spectral distance sits
// within fp32 rounding of a decision boundary.
__global__ __launch_bounds__(256) void match_pairs_kernel(const float* __restrict__ ov, const float* __restrict__ su,
                                                          const float* __restrict__ wn, const float* __restrict__ sn,
                                                          const int* __restrict__ pair_o, const int* __restrict__ pair_s,
                                                          int n_pairs, int We, long long* __restrict__ orientation,
                                                          float* __restrict__ distance, float* __restrict__ score) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pr = blockIdx.x * 4 + wave;
    if (pr >= n_pairs) return;                       // wave-uniform; the kernel has no barrier
    const int o = pair_o[pr], s = pair_s[pr];
    const int l31 = lane & 31, hk = lane >> 5;
    const int Wp = (We + 1) & ~1;
    const float* orow = ov + (size_t)o * 4096;
    const float* srow = su + (size_t)s * 64 * We;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    for (int r = 0; r < 64; ++r, orow += 64, srow += We) {
        for (int k = 0; k < Wp; k += 2) {
            const int kk = k + hk;                   // lanes 0-31 carry k, lanes 32-63 k + 1 (the A / B layout of 32x32x2)
            const float a = kk < We ? srow[kk] : 0.f;
            const float b0 = orow[(kk + l31) & 63];
            const float b1 = orow[(kk + l31 + 32) & 63];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
    }
    float v = acc0[0];
    int idx = l31;
    const float v1 = acc1[0];
    if (v1 > v) { v = v1; idx = 32 + l31; }
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const float vo = __shfl_xor(v, d, 64);
        const int io = __shfl_xor(idx, d, 64);
        if (vo > v || (vo == v && io < idx)) { v = vo; idx = io; }
    }
    if (lane == 0) {
        if (orientation) orientation[pr] = idx;
        if (score) score[pr] = v;
        if (distance) distance[pr] = 2.f * (1.f - v / (wn[(size_t)o * 64 + idx] * sn[s]));
    }
}

// The same list on the VECTOR pipe (round 6), still bit-identical: v_mfma_f32_32x32x2_f32 accumulates exactly as a chain of fused
// multiply-adds in k order (MI355X_MICROARCH.md, Matrix cores: 'exact f32 (= fmaf chain, bitwise)'), so lane j of ONE wave can run
// the pair's shift j as 64 x Wp v_fma_f32 -- acc = fma(su[r][k], ov[r][(k + j) & 63], acc), k padded to even with a zero
// multiplier exactly as the MFMA's second k lane is -- instead of all 32 rows of an MFMA tile carrying the same surface: 4,096
// FMAs per pair instead of 8,192 MFMAs of 64 cycles (one pair kept a matrix pipe busy for 110 us; the retrieval pass re-scores
// ~43 k pairs: 11 ms of its 233). The overhead row is written twice over into a wave-private LDS strip (128 floats, double
// buffered by row parity; a wave's LDS operations execute in order, so no barrier) and lane j reads strip[k + j] -- consecutive
// lanes, consecutive banks; the multiplier is wave-uniform (scalar loads). Arg-max and distance: the MFMA kernel's own code.
template <int WE>      // WE > 0: the embedding width as a constant (64: retrieval at fov 360); 0: any width
__global__ __launch_bounds__(256) void match_pairs_valu_kernel(const float* __restrict__ ov, const float* __restrict__ su,
                                                               const float* __restrict__ wn, const float* __restrict__ sn,
                                                               const int* __restrict__ pair_o, const int* __restrict__ pair_s,
                                                               int n_pairs, int We_rt, long long* __restrict__ orientation,
                                                               float* __restrict__ distance, float* __restrict__ score,
                                                               const int* __restrict__ n_dev, const float* __restrict__ thr,
                                                               int* __restrict__ counts) {
    __shared__ float strip[4][2][128];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pr = blockIdx.x * 4 + wave;
    // n_dev: the list's length lives on the device (witw_match_pairs_count: the list was written by witw_rank_count_band in the
    // same stream and nobody has read its length back); n_pairs is then the list's capacity
    if (n_dev != nullptr) n_pairs = min(n_pairs, __builtin_amdgcn_readfirstlane(*n_dev));
    if (pr >= n_pairs) return;                       // wave-uniform; the kernel has no barrier
    const int o = __builtin_amdgcn_readfirstlane(pair_o[pr]), s = __builtin_amdgcn_readfirstlane(pair_s[pr]);
    const int We = WE ? WE : We_rt;
    const int Wp = (We + 1) & ~1;
    const float* orow = ov + (size_t)o * 4096;
    const float* srow = su + (size_t)s * 64 * We;
    float acc = 0.f;
    float x = orow[lane];
    for (int r = 0; r < 64; ++r, srow += We) {
        float* buf = strip[wave][r & 1];
        buf[lane] = x;
        buf[64 + lane] = x;
        // The reads below take what OTHER lanes stored: per thread they never touch an address this thread wrote, so without a
        // compiler barrier the stores may sink below them (hipcc did exactly that to the peeled last row). The hardware needs
        // nothing: a wave's LDS operations execute in order.
        asm volatile("" ::: "memory");
        if (r + 1 < 64) x = orow[(r + 1) * 64 + lane];          // the next row, fetched under this row's chain
        const float* b = buf + lane;
        if (WE) {
#pragma unroll
            for (int k = 0; k < ((WE + 1) & ~1); ++k) acc = __builtin_fmaf(k < WE ? srow[k] : 0.f, b[k], acc);
        } else {
            for (int k = 0; k < Wp; ++k) acc = __builtin_fmaf(k < We ? srow[k] : 0.f, b[k], acc);
        }
    }
    // lanes 0-31: shift l31 against shift l31 + 32 first, then the butterfly over the 32 columns -- the comparison tree of the
    // MFMA kernel (acc0 / acc1), so that ties and NaNs resolve alike
    const int l31 = lane & 31;
    float v = acc;
    int idx = l31;
    const float v1 = __shfl_xor(acc, 32, 64);
    if (v1 > v) { v = v1; idx = 32 + l31; }
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const float vo = __shfl_xor(v, d, 64);
        const int io = __shfl_xor(idx, d, 64);
        if (vo > v || (vo == v && io < idx)) { v = vo; idx = io; }
    }
    if (lane == 0) {
        const float d = 2.f * (1.f - v / (wn[(size_t)o * 64 + idx] * sn[s]));
        if (orientation) orientation[pr] = idx;
        if (score) score[pr] = v;
        if (distance) distance[pr] = d;
        if (counts != nullptr && d <= thr[s]) atomicAdd(&counts[s], 1);      // the rank count's band: this row is at or below the threshold
    }
}

// Rank counting against a threshold with a rounding band: D holds distances known to eps (the spectral pass). Rows surely
// below the threshold are counted, rows inside [thr - eps, thr + eps] are appended to a pair list for exact re-scoring
// (n_pairs counts them all, also beyond the capacity: the caller then repeats with a larger list).
__global__ __launch_bounds__(256) void rank_band_kernel(const float* __restrict__ D, const float* __restrict__ thr, float eps,
                                                        int* __restrict__ counts, int* __restrict__ pair_o,
                                                        int* __restrict__ pair_s, int* __restrict__ n_pairs, int capacity,
                                                        int Bo, int Bs, int rows_per_block) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Bs) return;
    const float t = thr[q];
    const float lo = t - eps, hi = t + eps;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(Bo, r0 + rows_per_block);
    int c = 0;
    auto look = [&](float d, int o) {
        if (d < lo) ++c;
        else if (d <= hi) {
            const int slot = atomicAdd(n_pairs, 1);
            if (slot < capacity) {
                pair_o[slot] = o;
                pair_s[slot] = q;
            }
        }
    };
    int o = r0;
    for (; o + 8 <= r1; o += 8) {      // eight independent loads in flight per lane (one at a time the scan ran at 3.3 TB/s)
        float d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = D[(size_t)(o + u) * Bs + q];
#pragma unroll
        for (int u = 0; u < 8; ++u) look(d[u], o + u);
    }
    for (; o < r1; ++o) look(D[(size_t)o * Bs + q], o);
    if (c) atomicAdd(&counts[q], c);
}

}  // namespace

static int g_match_pairs_impl = 1;      // witw_match_pairs_impl

extern "C" {

long long witw_match_workspace_floats(int Bo, int Bs) { return (long long)Bo * 64 + Bs; }

int witw_match_fwd(const float* ov, const float* su, int Bo, int Bs, int We, long long* orientation, float* distance,
                   float* score, float* workspace, void* stream) {
    WITW_CHECK_ARG(ov && su && workspace, "match_fwd: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0, "match_fwd: empty batch Bo=%d Bs=%d", Bo, Bs);
    WITW_CHECK_ARG(We >= 1 && We <= 64, "match_fwd: surface embedding width %d outside [1,64]", We);
    hipStream_t st = (hipStream_t)stream;
    float* wn = workspace;
    float* sn = workspace + (size_t)Bo * 64;
    hipLaunchKernelGGL(window_norm_kernel, dim3(Bo), dim3(256), 0, st, ov, wn, We);
    hipLaunchKernelGGL(row_norm_kernel, dim3(Bs), dim3(256), 0, st, su, sn, 64 * We);
    MatchArgs a;
    a.ov = ov; a.su = su; a.wn = wn; a.sn = sn;
    a.orientation = orientation; a.distance = distance; a.score = score;
    a.Bo = Bo; a.Bs = Bs; a.We = We;
    const int gx = cdiv(Bs, MS);
    // small problems: 2 overheads per block (more blocks); large: 4 per block (less staging per FLOP)
    const char* force = getenv("WITW_MATCH_GENERIC");       // A/B aid: 1 = never use the pipelined kernels
    const bool pipelined = (long long)gx * cdiv(Bo, 4) >= 256 && !(force && atoi(force) != 0);
    if (pipelined && We >= 63) {
        hipLaunchKernelGGL(match_kernel_w64, dim3(gx, cdiv(Bo, 4)), dim3(NT), 0, st, a);
    } else if (pipelined && We <= 16) {
        hipLaunchKernelGGL((match_kernel_rows<4>), dim3(gx, cdiv(Bo, 4)), dim3(NT), 0, st, a);
    } else if (pipelined && We <= 32) {
        hipLaunchKernelGGL((match_kernel_rows<2>), dim3(gx, cdiv(Bo, 4)), dim3(NT), 0, st, a);
    } else if (pipelined) {      // 33..62 columns: one row per stage, zero-padded to 64
        hipLaunchKernelGGL((match_kernel_rows<1>), dim3(gx, cdiv(Bo, 4)), dim3(NT), 0, st, a);
    } else if ((long long)gx * cdiv(Bo, 4) >= 512) {
        hipLaunchKernelGGL((match_kernel<2, 2>), dim3(gx, cdiv(Bo, 4)), dim3(NT), 0, st, a);
    } else if ((long long)gx * cdiv(Bo, 2) >= 256) {
        hipLaunchKernelGGL((match_kernel<1, 2>), dim3(gx, cdiv(Bo, 2)), dim3(NT), 0, st, a);
    } else if ((long long)cdiv(Bs, 64) * cdiv(Bo, 2) >= 96) {      // a single minibatch (128 x 128): one overhead per workgroup, the
        const dim3 g2(cdiv(Bs, 64), Bo);                                       // waves split the shifts: a wave on every SIMD
        constexpr int RPC = NSPLIT_RPC;                                        // embedding rows per LDS stage and barrier
        constexpr size_t lds = 2 * (64 * (RPC * 64 + 1) + RPC * 128) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set && lds > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(match_kernel_nsplit<16, RPC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(match_kernel_nsplit<32, RPC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(match_kernel_nsplit<64, RPC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (We <= 16) hipLaunchKernelGGL((match_kernel_nsplit<16, RPC>), g2, dim3(NT), lds, st, a);
        else if (We <= 32) hipLaunchKernelGGL((match_kernel_nsplit<32, RPC>), g2, dim3(NT), lds, st, a);
        else hipLaunchKernelGGL((match_kernel_nsplit<64, RPC>), g2, dim3(NT), lds, st, a);
    } else {        // smaller still: 64-surface blocks of two overheads
        hipLaunchKernelGGL((match_kernel<1, 1>), dim3(cdiv(Bs, 64), cdiv(Bo, 2)), dim3(NT), 0, st, a);
    }
    WITW_CHECK_LAUNCH("match_fwd");
    return WITW_OK;
}

// grad_distance [Bo,Bs] -> grad_ov [Bo,16,4,64], grad_su [Bs,16,4,We]. orientation / score / workspace are
// the outputs of witw_match_fwd on the same (ov, su) (workspace must not have been overwritten).
// Overhead splits of the surface-gradient kernel: enough (surface, split) blocks to fill the chip when the batch has few
// surfaces (one block per surface otherwise), at least 32 overheads per split.
static int match_bwd_splits(int Bo, int Bs) {
    int splits = cdiv(768, Bs);
    if (splits > cdiv(Bo, 32)) splits = cdiv(Bo, 32);
    return splits < 1 ? 1 : splits;
}

// floats of scratch witw_match_bwd wants for grad_su (0: none needed)
long long witw_match_bwd_scratch_floats(int Bo, int Bs, int We) {
    if (Bo <= 0 || Bs <= 0 || We < 1 || We > 64) return -1;
    const int splits = match_bwd_splits(Bo, Bs);
    return splits > 1 ? (long long)splits * Bs * (64 * We + 1) : 0;
}

int witw_match_bwd(const float* ov, const float* su, const long long* orientation, const float* score, const float* workspace,
                   const float* grad_distance, float* grad_ov, float* grad_su, float* scratch, int Bo, int Bs, int We,
                   void* stream) {
    WITW_CHECK_ARG(ov && su && orientation && score && workspace && grad_distance, "match_bwd: null pointer");
    WITW_CHECK_ARG(grad_ov || grad_su, "match_bwd: no output requested");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0 && We >= 1 && We <= 64, "match_bwd: bad shape Bo=%d Bs=%d We=%d", Bo, Bs, We);
    hipStream_t st = (hipStream_t)stream;
    const float* wn = workspace;
    const float* sn = workspace + (size_t)Bo * 64;
    if (grad_su) {
        const int splits = scratch ? match_bwd_splits(Bo, Bs) : 1;      // without scratch: one block per surface
        const int ops = cdiv(Bo, splits);
        hipLaunchKernelGGL(match_bwd_su_kernel, dim3(Bs, splits), dim3(256), 0, st, ov, su, orientation, score, wn, sn,
                           grad_distance, grad_su, Bo, Bs, We, ops, scratch);
        if (splits > 1) {
            const size_t n = (size_t)Bs * 64 * We;
            hipLaunchKernelGGL(match_bwd_su_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, su, scratch, grad_su,
                               Bs, 64 * We, splits);
        }
    }
    if (grad_ov)
        hipLaunchKernelGGL(match_bwd_ov_kernel, dim3(Bo), dim3(256), 0, st, ov, su, orientation, score, wn, sn, grad_distance,
                           grad_ov, Bo, Bs, We);
    WITW_CHECK_LAUNCH("match_bwd");
    return WITW_OK;
}

int witw_crop_overhead(const float* ov, const long long* orientation, float* out, int Bo, int Bs, int We, void* stream) {
    WITW_CHECK_ARG(ov && orientation && out, "crop_overhead: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0 && We >= 1 && We <= 64, "crop_overhead: bad shape Bo=%d Bs=%d We=%d", Bo, Bs, We);
    const size_t total = (size_t)Bo * Bs * 64 * We;
    WITW_CHECK_ARG((total + 255) / 256 <= 0x7fffffffULL, "crop_overhead: output too large");
    hipLaunchKernelGGL(crop_overhead_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ov,
                       orientation, out, Bo, Bs, We, total);
    WITW_CHECK_LAUNCH("crop_overhead");
    return WITW_OK;
}

int witw_l2_distance(const float* cropped, const float* su, float* distance, int Bo, int Bs, int n, void* stream) {
    WITW_CHECK_ARG(cropped && su && distance, "l2_distance: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0 && n > 0, "l2_distance: bad shape Bo=%d Bs=%d n=%d", Bo, Bs, n);
    WITW_CHECK_ARG((long long)Bo * Bs <= 0x7fffffffLL, "l2_distance: too many pairs");
    hipLaunchKernelGGL(l2_distance_kernel, dim3(Bo * Bs), dim3(256), 0, (hipStream_t)stream, cropped, su, distance, Bs, n);
    WITW_CHECK_LAUNCH("l2_distance");
    return WITW_OK;
}

int witw_rank_count(const float* distance, int* ranks, int Bo, int Bs, int true_offset, void* stream) {
    WITW_CHECK_ARG(distance && ranks, "rank_count: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0, "rank_count: bad shape Bo=%d Bs=%d", Bo, Bs);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv(Bs, 256)), dim3(256), 0, st, ranks, Bs);
    const int rows = 512;
    hipLaunchKernelGGL(rank_count_kernel, dim3(cdiv(Bs, 256), cdiv(Bo, rows)), dim3(256), 0, st, distance, ranks, Bo, Bs,
                       true_offset, rows);
    WITW_CHECK_LAUNCH("rank_count");
    return WITW_OK;
}

int witw_rank_count_thresh(const float* distance, const float* threshold, int* ranks, int Bo, int Bs, void* stream) {
    WITW_CHECK_ARG(distance && threshold && ranks, "rank_count_thresh: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0, "rank_count_thresh: bad shape Bo=%d Bs=%d", Bo, Bs);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv(Bs, 256)), dim3(256), 0, st, ranks, Bs);
    const int rows = 512;
    hipLaunchKernelGGL(rank_count_thresh_kernel, dim3(cdiv(Bs, 256), cdiv(Bo, rows)), dim3(256), 0, st, distance, threshold,
                       ranks, Bo, Bs, rows);
    WITW_CHECK_LAUNCH("rank_count_thresh");
    return WITW_OK;
}

// n_pairs (overhead row, surface row) pairs -> orientation / distance / score [n_pairs] (any may be null), bit-identical to
// the entries witw_match_fwd writes for those pairs. wn [Bo,64] / sn [Bs]: the norms of a witw_match_fwd or
// witw_match_fwd_dft workspace over the same ov / su (window norms first, surface norms behind them).
int witw_match_pairs(const float* ov, const float* su, const float* wn, const float* sn, const int* pair_o, const int* pair_s,
                     int n_pairs, int Bo, int Bs, int We, long long* orientation, float* distance, float* score, void* stream) {
    WITW_CHECK_ARG(ov && su && wn && sn && pair_o && pair_s, "match_pairs: null pointer");
    WITW_CHECK_ARG(n_pairs > 0 && Bo > 0 && Bs > 0, "match_pairs: empty list n=%d Bo=%d Bs=%d", n_pairs, Bo, Bs);
    WITW_CHECK_ARG(We >= 1 && We <= 64, "match_pairs: surface embedding width %d outside [1,64]", We);
    if (g_match_pairs_impl == 0)
        hipLaunchKernelGGL(match_pairs_kernel, dim3(cdiv(n_pairs, 4)), dim3(256), 0, (hipStream_t)stream, ov, su, wn, sn, pair_o, pair_s,
                           n_pairs, We, orientation, distance, score);
    else if (We == 64)
        hipLaunchKernelGGL((match_pairs_valu_kernel<64>), dim3(cdiv(n_pairs, 4)), dim3(256), 0, (hipStream_t)stream, ov, su, wn, sn, pair_o,
                           pair_s, n_pairs, We, orientation, distance, score, (const int*)nullptr, (const float*)nullptr, (int*)nullptr);
    else
        hipLaunchKernelGGL((match_pairs_valu_kernel<0>), dim3(cdiv(n_pairs, 4)), dim3(256), 0, (hipStream_t)stream, ov, su, wn, sn, pair_o,
                           pair_s, n_pairs, We, orientation, distance, score, (const int*)nullptr, (const float*)nullptr, (int*)nullptr);
    WITW_CHECK_LAUNCH("match_pairs");
    return WITW_OK;
}

// The band of witw_rank_count_band resolved WITHOUT a host round trip: the first min(*n_pairs_dev, capacity) pairs of the list are
// re-scored exactly (as witw_match_pairs) and counts[pair_s[i]] is incremented for every pair whose exact distance is <=
// threshold[pair_s[i]] -- after it counts[q] = #{o : D_exact[o][q] <= threshold[q]} provided the list did not overflow (the caller
// checks *n_pairs_dev <= capacity once, at the end of its pass). Launches capacity / 4 workgroups; those beyond the list exit.
int witw_match_pairs_count(const float* ov, const float* su, const float* wn, const float* sn, const int* pair_o, const int* pair_s,
                           const int* n_pairs_dev, int capacity, int Bo, int Bs, int We, const float* threshold, int* counts, void* stream) {
    WITW_CHECK_ARG(ov && su && wn && sn && pair_o && pair_s && n_pairs_dev && threshold && counts, "match_pairs_count: null pointer");
    WITW_CHECK_ARG(capacity > 0 && Bo > 0 && Bs > 0, "match_pairs_count: bad arguments capacity=%d Bo=%d Bs=%d", capacity, Bo, Bs);
    WITW_CHECK_ARG(We >= 1 && We <= 64, "match_pairs_count: surface embedding width %d outside [1,64]", We);
    if (We == 64)
        hipLaunchKernelGGL((match_pairs_valu_kernel<64>), dim3(cdiv(capacity, 4)), dim3(256), 0, (hipStream_t)stream, ov, su, wn, sn, pair_o,
                           pair_s, capacity, We, (long long*)nullptr, (float*)nullptr, (float*)nullptr, n_pairs_dev, threshold, counts);
    else
        hipLaunchKernelGGL((match_pairs_valu_kernel<0>), dim3(cdiv(capacity, 4)), dim3(256), 0, (hipStream_t)stream, ov, su, wn, sn, pair_o,
                           pair_s, capacity, We, (long long*)nullptr, (float*)nullptr, (float*)nullptr, n_pairs_dev, threshold, counts);
    WITW_CHECK_LAUNCH("match_pairs_count");
    return WITW_OK;
}

// Which kernel witw_match_pairs runs: 1 = the v_fma_f32 chain on the vector pipe (default), 0 = the v_mfma_f32_32x32x2_f32 chain of
// rounds 4-5 (one wave per pair, 31/32 of the matrix work redundant). Both give the bits of witw_match_fwd. impl < 0 only queries.
// Returns the previous setting.
int witw_match_pairs_impl(int impl) {
    const int prev = g_match_pairs_impl;
    if (impl >= 0) g_match_pairs_impl = impl ? 1 : 0;
    return prev;
}

// counts[q] = #{o : D[o,q] < threshold[q] - eps}; the (o, q) with |D[o,q] - threshold[q]| <= eps go to pair_o / pair_s
// (first `capacity` of them) and *n_pairs = how many there were. counts and *n_pairs are zeroed here.
int witw_rank_count_band(const float* distance, const float* threshold, float eps, int* counts, int* pair_o, int* pair_s,
                         int* n_pairs, int capacity, int Bo, int Bs, void* stream) {
    WITW_CHECK_ARG(distance && threshold && counts && pair_o && pair_s && n_pairs, "rank_count_band: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0 && capacity > 0 && eps >= 0.f, "rank_count_band: bad arguments Bo=%d Bs=%d capacity=%d", Bo, Bs, capacity);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv(Bs, 256)), dim3(256), 0, st, counts, Bs);
    hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(256), 0, st, n_pairs, 1);
    const int rows = 512;
    hipLaunchKernelGGL(rank_band_kernel, dim3(cdiv(Bs, 256), cdiv(Bo, rows)), dim3(256), 0, st, distance, threshold, eps, counts,
                       pair_o, pair_s, n_pairs, capacity, Bo, Bs, rows);
    WITW_CHECK_LAUNCH("rank_count_band");
    return WITW_OK;
}

}  // extern "C"

// ---- top-k smallest distances per query (retrieval indices, BASELINE config C5 / north_star "top-k").
// D is [Bo][Bs] (gallery rows x queries). A block covers 64 consecutive queries (lanes -> coalesced 256-B
// row reads) and its 4 waves split the gallery rows; every thread keeps a sorted K-list in registers, the 4
// lists of a query are merged through LDS. Order: (distance, gallery index) ascending, so ties resolve to the
// lower index and the result is deterministic and independent of the launch shape.
namespace {

// Gallery rows can be split over blockIdx.y (pv / pi non-null): a block then ranks rows [y*rps, (y+1)*rps) and leaves its k
// candidates per query in the workspace, merged by topk_merge_kernel -- one block per 64 queries walking 125,000 rows alone is
// 64 workgroups of dependent loads (16 ms per 125,000 x 4,096 launch; the matrix streams from HBM in 0.5 ms).
template <int K>
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ D, float* __restrict__ vals,
                                                   long long* __restrict__ idx, int Bo, int Bs, int k, long long row_offset,
                                                   int rps, float* __restrict__ pv, int* __restrict__ pi,
                                                   const float* __restrict__ tau, int tau_stride) {
    __shared__ float sv[4][K][64];
    __shared__ int si[4][K][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rps, r1 = min(Bo, r0 + rps);
    float bv[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bv[j] = __builtin_inff();
        bi[j] = 0x7fffffff;
    }
    // tau (optional): per query an UPPER BOUND of its k-th smallest distance -- the k-th smallest of the gallery's first rows, from a
    // pass over those alone. Rows above it cannot be among the k best (ties at the bound still can) and are dropped by one compare:
    // without it every split starts from empty lists and nearly every batch of its first thousand rows triggers an insertion
    // (1.3 TB/s over the 2 GB matrix); with it the scan is loads and compares.
    const float tq = (tau != nullptr && q < Bs) ? tau[(size_t)q * tau_stride] : __builtin_inff();
    auto offer = [&](float d, int o) {
        if (d != d) d = __builtin_inff();          // NaN sorts last
        if (d <= tq && (d < bv[K - 1] || (d == bv[K - 1] && o < bi[K - 1]))) {
            float cv = d;
            int ci = o;
#pragma unroll
            for (int j = 0; j < K; ++j) {          // insertion into the sorted list
                const bool lt = cv < bv[j] || (cv == bv[j] && ci < bi[j]);
                const float tv = lt ? bv[j] : cv;
                const int ti = lt ? bi[j] : ci;
                bv[j] = lt ? cv : bv[j];
                bi[j] = lt ? ci : bi[j];
                cv = tv;
                ci = ti;
            }
        }
    };
    if (q < Bs) {
        int o = r0 + wave;
        // eight independent loads per batch, and the NEXT batch requested before this one is ranked: with the loads of one batch
        // only, a wave's memory round trip and its ranking alternate and the scan ran at 1.5 TB/s of the 2 GB matrix
        float dn[8];
        if (o + 28 < r1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) dn[u] = D[(size_t)(o + 4 * u) * Bs + q];
        }
        for (; o + 28 < r1; o += 32) {
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = dn[u];
            if (o + 32 + 28 < r1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) dn[u] = D[(size_t)(o + 32 + 4 * u) * Bs + q];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (d[u] != d[u]) d[u] = __builtin_inff();
            // A wave executes an insertion (16 compare-exchange steps) whenever ANY of its 64 queries takes a row; offered one by
            // one, nearly every row of the first few thousand triggers one (64 K / n per row at row n of a query's stream). Per
            // batch of eight rows each lane instead inserts its SMALLEST admissible candidate, and the wave repeats only while some
            // lane still holds another one: one or two insertions per batch instead of eight (the list is a set ordered by
            // (distance, row): the order of insertion does not matter).
            unsigned live = 0xffu;
            for (;;) {
                float cv = __builtin_inff();
                int cu = -1;
#pragma unroll
                for (int u = 0; u < 8; ++u) {      // smallest admissible (distance, row) among the rows not yet dealt with
                    const bool adm = ((live >> u) & 1u) && d[u] <= tq && (d[u] < bv[K - 1] || (d[u] == bv[K - 1] && o + 4 * u < bi[K - 1]));
                    if (!adm) live &= ~(1u << u);
                    if (adm && (cu < 0 || d[u] < cv)) { cv = d[u]; cu = u; }      // equal distances: the lower row (lower u) stays
                }
                if (!__builtin_amdgcn_ballot_w64(cu >= 0)) break;
                if (cu >= 0) {
                    live &= ~(1u << cu);
                    float iv = cv;
                    int ci = o + 4 * cu;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const bool lt = iv < bv[j] || (iv == bv[j] && ci < bi[j]);
                        const float tv = lt ? bv[j] : iv;
                        const int ti = lt ? bi[j] : ci;
                        bv[j] = lt ? iv : bv[j];
                        bi[j] = lt ? ci : bi[j];
                        iv = tv;
                        ci = ti;
                    }
                }
            }
        }
        for (; o < r1; o += 4) offer(D[(size_t)o * Bs + q], o);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        sv[wave][j][lane] = bv[j];
        si[wave][j][lane] = bi[j];
    }
    __syncthreads();
    if (wave == 0 && q < Bs) {                        // 4-way merge of sorted lists, one query per lane
        int h[4] = {0, 0, 0, 0};
        for (int out = 0; out < k; ++out) {
            int best = -1;
            float v = __builtin_inff();
            int ix = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (h[w] < K) {
                    const float cv = sv[w][h[w]][lane];
                    const int ci = si[w][h[w]][lane];
                    if (best < 0 || cv < v || (cv == v && ci < ix)) { best = w; v = cv; ix = ci; }
                }
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) h[w] += (w == best) ? 1 : 0;
            if (pv != nullptr) {
                pv[((size_t)blockIdx.y * k + out) * Bs + q] = v;
                pi[((size_t)blockIdx.y * k + out) * Bs + q] = ix;
            } else {
                vals[(size_t)q * k + out] = v;
                idx[(size_t)q * k + out] = (ix == 0x7fffffff) ? -1 : (long long)ix + row_offset;
            }
        }
    }
}

// splits x k sorted candidates per query -> the k best by (distance, row); one thread per query, coalesced candidate reads
template <int K>
__global__ __launch_bounds__(64) void topk_merge_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                       float* __restrict__ vals, long long* __restrict__ idx, int Bs, int k,
                                                       int splits, long long row_offset) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= Bs) return;
    float bv[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bv[j] = __builtin_inff();
        bi[j] = 0x7fffffff;
    }
    auto offer = [&](float cv, int ci) {
        if (cv < bv[K - 1] || (cv == bv[K - 1] && ci < bi[K - 1])) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const bool lt = cv < bv[j] || (cv == bv[j] && ci < bi[j]);
                const float tv = lt ? bv[j] : cv;
                const int ti = lt ? bi[j] : ci;
                bv[j] = lt ? cv : bv[j];
                bi[j] = lt ? ci : bi[j];
                cv = tv;
                ci = ti;
            }
        }
    };
    const int n = splits * k;
    int c = 0;
    for (; c + 8 <= n; c += 8) {      // eight candidates in flight (one thread per query: a chain of dependent loads otherwise)
        float cv[8];
        int ci[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cv[u] = pv[(size_t)(c + u) * Bs + q];
            ci[u] = pi[(size_t)(c + u) * Bs + q];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) offer(cv[u], ci[u]);
    }
    for (; c < n; ++c) offer(pv[(size_t)c * Bs + q], pi[(size_t)c * Bs + q]);
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k) {
            vals[(size_t)q * k + j] = bv[j];
            idx[(size_t)q * k + j] = (bi[j] == 0x7fffffff) ? -1 : (long long)bi[j] + row_offset;
        }
}

}  // namespace

constexpr int TOPK_SAMPLE = 1024;      // rows of the threshold pass of long galleries (witw_topk_smallest_ws)
constexpr int TOPK_SAMPLE_SPLITS = 8;  // ... ranked in this many row splits

static int topk_splits(int Bo, int Bs) {
    // enough (query tile, split) workgroups to fill the chip several times over, at least 512 rows per split
    int s = cdiv(2048, cdiv(Bs, 64));
    if (s > cdiv(Bo, 512)) s = cdiv(Bo, 512);
    return s < 1 ? 1 : s;
}

template <int K>
static void topk_launch(const float* D, float* values, long long* indices, int Bo, int Bs, int k, long long row_offset,
                        void* workspace, hipStream_t st) {
    const int splits = workspace ? topk_splits(Bo, Bs) : 1;
    if (splits <= 1) {
        hipLaunchKernelGGL((topk_kernel<K>), dim3(cdiv(Bs, 64)), dim3(256), 0, st, D, values, indices, Bo, Bs, k, row_offset, Bo,
                           (float*)nullptr, (int*)nullptr, (const float*)nullptr, 0);
        return;
    }
    float* pv = (float*)workspace;
    int* pi = (int*)(pv + (size_t)splits * k * Bs);
    const int rps = cdiv(Bo, splits);
    // long galleries: the k best of the first TOPK_SAMPLE rows first (into `values`, which the merge overwrites at the end); their k-th
    // distance bounds every query's k-th distance over all rows and is the main pass's admission threshold
    const float* tau = nullptr;
    if (Bo >= 16 * TOPK_SAMPLE && splits >= TOPK_SAMPLE_SPLITS) {      // (the sample is ranked in row splits too: the workspace is free until the main pass)
        hipLaunchKernelGGL((topk_kernel<K>), dim3(cdiv(Bs, 64), TOPK_SAMPLE_SPLITS), dim3(256), 0, st, D, values, indices, TOPK_SAMPLE, Bs, k,
                           row_offset, TOPK_SAMPLE / TOPK_SAMPLE_SPLITS, pv, pi, (const float*)nullptr, 0);
        hipLaunchKernelGGL((topk_merge_kernel<K>), dim3(cdiv(Bs, 64)), dim3(64), 0, st, pv, pi, values, indices, Bs, k, TOPK_SAMPLE_SPLITS,
                           row_offset);
        tau = values + (k - 1);
    }
    hipLaunchKernelGGL((topk_kernel<K>), dim3(cdiv(Bs, 64), splits), dim3(256), 0, st, D, values, indices, Bo, Bs, k, row_offset,
                       rps, pv, pi, tau, k);
    hipLaunchKernelGGL((topk_merge_kernel<K>), dim3(cdiv(Bs, 64)), dim3(64), 0, st, pv, pi, values, indices, Bs, k, splits, row_offset);
}

// bytes of workspace that let witw_topk_smallest_ws split the gallery rows over workgroups (0: a single pass is used anyway)
extern "C" long long witw_topk_workspace_bytes(int Bo, int Bs, int k) {
    if (Bo <= 0 || Bs <= 0 || k < 1 || k > 32) return -1;
    const int splits = topk_splits(Bo, Bs);
    return splits > 1 ? (long long)splits * k * Bs * 8 : 0;
}

extern "C" int witw_topk_smallest_ws(const float* distance, float* values, long long* indices, int Bo, int Bs, int k,
                                     long long row_offset, void* workspace, void* stream) {
    WITW_CHECK_ARG(distance && values && indices, "topk_smallest: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0, "topk_smallest: bad shape Bo=%d Bs=%d", Bo, Bs);
    WITW_CHECK_ARG(k >= 1 && k <= 32, "topk_smallest: k=%d outside [1,32]", k);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 8) topk_launch<8>(distance, values, indices, Bo, Bs, k, row_offset, workspace, st);
    else if (k <= 16) topk_launch<16>(distance, values, indices, Bo, Bs, k, row_offset, workspace, st);
    else topk_launch<32>(distance, values, indices, Bo, Bs, k, row_offset, workspace, st);
    WITW_CHECK_LAUNCH("topk_smallest");
    return WITW_OK;
}

extern "C" int witw_topk_smallest(const float* distance, float* values, long long* indices, int Bo, int Bs, int k,
                                  long long row_offset, void* stream) {
    return witw_topk_smallest_ws(distance, values, indices, Bo, Bs, k, row_offset, nullptr, stream);
}

