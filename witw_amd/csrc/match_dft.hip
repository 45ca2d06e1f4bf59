// Orientation search + chord distance through the row spectra (gfx950, fp32 MFMA): the retrieval form of the fused match
// (BASELINE config C5; model/cvig_fov.py:297-363, called at :547-549 for every (gallery row, query) pair).
//
// score[o,s,shift] = sum_{ch,k} ov[o,ch,(k+shift)%64] * su[s,ch,k] is a circular cross-correlation along the 64 columns, summed
// over the 64 (channel,row) lines. With X_f = sum_k x[k] e^{-2 pi i f k/64} per line (su zero-padded to 64 columns):
//   C_f[o,s]   = sum_ch OV_f[o,ch] * conj(SU_f[s,ch])                                  f = 0..32
//   score[o,s,shift] = (1/64) [ C_0 + (-1)^shift C_32 + 2 sum_{f=1..31} (Re C_f cos(2 pi f shift/64) - Im C_f sin(..)) ]
// 2*2*128*33 + 2*2*32*33 = 21k FLOP per pair instead of 2*64*4096 = 524k of the direct form (match.hip), both on the fp32 MFMA
// (the algorithmic count; the kernel runs 32 slots: the real spectra at f = 0 and f = 32 share one, see NSLOT).
//
// Kernel: persistent workgroups (one per CU), a tile = 32 surfaces x 32 overheads, 4 waves = 2 surface teams x {even, odd}
// frequencies. Per frequency slot
//   GEMM 1 (32x32x2 f32 MFMA, K = 128 = 64 lines x {re,im}): rows = 16 surfaces x {Re C, Im C}, columns = 32 overheads. Rows are
//          ordered so that accumulator register r of lanes 0-31 holds Re C[surface r] and of lanes 32-63 Im C[surface r] of the
//          same (surface, overhead): exactly the A operand (32 overheads x K=2) of
//   GEMM 2 (one 32x32x2 MFMA per surface and slot): [32 overheads x (Re,Im)] x [(cos,-sin) x 32 shifts], accumulated over the
//          slots into 16 x f32x16 registers per wave (all 256 accumulation registers of the wave).
//          Round 5: the coefficients are the A operand and C the B operand, so the accumulator holds [32 shifts x 32 overheads]:
//          a lane owns ONE overhead and 16 of the 32 shifts in its 16 registers (the other 16 sit in lane ^ 32).
// Even frequencies give E[shift], odd ones O[shift] for shift < 32; score[shift] = E + O, score[shift+32] = E - O. The two
// waves of a team exchange HALF of their tiles register for register through LDS (the E wave finishes surfaces 0-7 of the
// team, the O wave 8-15: 16-byte writes and reads, no transposition), every lane scans its 16 shifts x {+, -} of a pair in
// registers, joins the other half-wave's result (first maximum wins, as torch.argmax) and writes
// orientation / score / distance like match.hip does.
#include "common.h"
#include "spectrum64_gen.h"

// WITW_DFT_DIAG (diagnostic builds, wrong results): 1 = no staging DMA inside the steps, 2 = no barrier / vmcnt wait per step,
// 4 = no operand reads inside the groups, 8 = the tile loop ends before the epilogue
#ifndef WITW_DFT_DIAG
#define WITW_DFT_DIAG 0
#endif
// WITW_DFT_PHASES=1 (diagnostic build): with WITW_DFT_STAMPS=2 in the environment the product instantiation sums s_memrealtime
// over the step loops and over the epilogues of a workgroup's tiles; printed to stderr
#ifndef WITW_DFT_PHASES
#define WITW_DFT_PHASES 0
#endif
namespace {

// Storage slots of a spectrum, [P(64 lines) | Q(64 lines)] each: slot t = 1..31 holds (Re, Im) of frequency t; the spectra at
// frequencies 0 and 32 are real, and slot 0 holds BOTH: P = X_0, Q = X_32 (round 5: 33 -> 32 slots, 17 -> 16 steps per tile, and
// no phantom 34th slot whose reads -- a neighbour's values times a zero coefficient -- let a NaN embedding reach the row before it)
constexpr int NSLOT = 32;
constexpr int SPEC = NSLOT * 128;      // floats per embedding spectrum (16 KB)
constexpr int NSTEP = 16;              // step i: even waves slot 2i, odd waves slot 2i+1
// LDS rows: 128 floats [P 64 | Q 64], unpadded (a 16-byte LDS-DMA writes 1 KB = two whole rows contiguously), with the 16-byte
// slots of each half XOR-swizzled by (row & 15): logical slot c of row r sits at slot c ^ (r & 15). ds_read_b64 of one k-group
// by 32 rows then touches every bank pair twice (rows r and r+16, and a surface's P and Q halves): 2-way, 4 LDS cycles.
constexpr int ROW_F = 128;
constexpr int A_F = 2 * 32 * ROW_F;    // [parity][32 surfaces]
constexpr int B_F = 2 * 32 * ROW_F;    // [parity][32 overheads]
constexpr int STAGE_F = A_F + B_F;     // 16384 floats
constexpr int LDS_F = 2 * STAGE_F;     // stage 0 | stage 1 / the epilogue exchange (4 waves x 16 KB per round)

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// global -> LDS, 16 B per lane, 1 KB of contiguous LDS per wave instruction at lds_addr (M0); out-of-range lanes write 0
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");
}

// One ds_read_b64 (the compiler would fuse neighbouring ones into ds_read2_b64, which is banked like ds_read_b32: 2-way on
// these rows). The results are ordered by lds_wait<N>() below, which also names them so that no MFMA moves above the wait.
template <int OFF>
__device__ __forceinline__ f32x2 lds_read64(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// operand address of k-group U: base ^ (U << 4) (the slot swizzle); formed in the GEMM-2 phase, see XA / XB / XC below
template <int IMM>
__device__ __forceinline__ unsigned lds_xor(unsigned addr) { return addr ^ (unsigned)IMM; }
// the step's inverse-transform coefficient, by hand as well (issued in group 14, covered by group 15's lgkmcnt(0)): a
// compiler-issued LDS read would be waited for with lgkmcnt(0) at its use, behind the next step's operand reads already in flight
__device__ __forceinline__ float lds_read32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x2& a, f32x2& b, f32x2& c, f32x2& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// GEMM-1 MFMA with its accumulator pinned to VGPRs. The wave owns 16 x 16 long-lived accumulation registers (all 256 AGPRs);
// left to the compiler the two short-lived GEMM-1 accumulators also go to AGPRs, 288 > 256, and one long-lived tile is shuttled
// through v_accvgpr moves every step: 79 instead of 69 cycles per MFMA in tools/mfma_rate.cpp's model of this loop. Inline asm
// is outside the compiler's hazard recogniser: the required wait states sit in mfma_settle() below.
__device__ __forceinline__ void mfma_v(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first MFMA of a chain: C = 0 as an inline constant instead of 16 zeroed registers
__device__ __forceinline__ void mfma_v0(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
// 16 accumulation registers back to zero for the next tile, as ONE instruction on the matrix pipe, which is idle during the
// epilogue (0 x 0 + 0), instead of 16 v_accvgpr_write on the vector pipe, which is the busy one there. asm: the builtin with
// constant operands is folded back into the 16 writes
__device__ __forceinline__ void mfma_zero(f32x16& acc, float zero) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, 0" : "=a"(acc) : "v"(zero));
}
// 16-pass XDL write -> VALU read of the result: 18 wait states (CDNA3 ISA, manually inserted wait states)
__device__ __forceinline__ void mfma_settle(f32x16& x, f32x16& y) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x), "+v"(y));
}

// One pair's 16 shifts of a lane (register q = shift (q & 3) + 8 (q >> 2) + 4 hk; the 4 hk is added by the caller) x {E + O at
// that shift, E - O at shift + 32}: the largest value, its index in torch.argmax order (smallest index among equal values) and,
// GAP, the runner-up value. max(E + O, E - O) = E + |O| and a negative O puts it at shift + 32 (O = -0 cannot come out of an
// MFMA sum that started at +0), so the scan looks at 16 values instead of 32. The index travels as (sign bit of O | shift),
// ordered like the arg-max index under an unsigned minimum; no comparison of O, whose VCC result would cost two wait states
// before the v_cndmask that reads it (gfx950).
constexpr unsigned NOKEY = 0xffffffffu;
__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }      // (min() resolves to the int overload)
// lo / hi = the lower / upper half-wave's x, in every lane: v_permlane32_swap exchanges the first operand's lanes 32-63 with the
// second one's lanes 0-31 (tools/debug/permlane_probe.cpp)
template <typename T>
__device__ __forceinline__ void half_swap(T x, T& lo, T& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned r0 = r[0], r1 = r[1];      // (a __builtin_bit_cast of the element expression r[1] itself reads element 0)
    lo = __builtin_bit_cast(T, r0);
    hi = __builtin_bit_cast(T, r1);
}
// VONLY: only the largest value is wanted (no orientation output and a full-width surface, whose window norm does not depend on
// the shift): the index search -- two thirds of the scan's instructions -- is left out
template <bool GAP, bool VONLY>
__device__ __forceinline__ void scan16(const f32x16& E, const f32x16& O, float& best, unsigned& key, float& second) {
    float m[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] = E[q] + fabsf(O[q]);
    float v = -INFINITY, s2 = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        // GAP: the second largest of the 32 values rides along (E - |O| never is the largest of its pair)
        if (GAP) s2 = fmaxf(fmaxf(s2, fminf(m[q], v)), E[q] - fabsf(O[q]));
        v = fmaxf(v, m[q]);
    }
    unsigned k = NOKEY;
#pragma unroll
    for (int q = 0; q < (VONLY ? 0 : 16); ++q) {
        const unsigned c = (unsigned)((q & 3) + 8 * (q >> 2));
        const float oq = O[q];      // (a __builtin_bit_cast of the element expression O[q] itself reads element 0 for every q)
        const unsigned kq = (__builtin_bit_cast(unsigned, oq) & 0x80000000u) | c;      // v_and + v_or (as asm v_bfi_b32: more moves)
        k = umin(k, m[q] == v ? kq : NOKEY);
    }
    best = v;
    key = k;
    if (GAP) second = s2;
}

struct DftArgs {
    const float* spec_ov;    // [Bo][32][128]
    const float* spec_su;    // [Bs][32][128]
    const float* dtab;       // [32][64]: lane (hk, shift) -> inverse-transform coefficient of the slot's (Re | Im) at that shift
    const float* wn;         // [Bo,64] window norms per shift
    const float* sn;         // [Bs]    surface norms
    long long* orientation;  // [Bo,Bs] or null
    float* distance;         // [Bo,Bs] or null
    float* score;            // [Bo,Bs] or null
    float* gap;              // [Bo,Bs] or null (GAP instantiation): best score - runner-up score over the 64 shifts
    int Bo, Bs, nbx, nby;
    unsigned long long* stamps;      // null, or 64 slots per 4096th workgroup (WITW_DFT_STAMPS=1: in-kernel timeline)
};

// REC: the diagnostic instantiation that records the in-kernel timeline (costs registers: the product launch uses REC = false)
// GAP: the shift scan also tracks the runner-up score and writes best - runner-up (narrow surfaces: the caller re-scores the
// pairs whose two best shifts tie to rounding, cvig_fov._dft_pass_narrow)
template <bool REC, bool GAP, bool VONLY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void match_dft_kernel(DftArgs p) {
    __shared__ __attribute__((aligned(1024))) float smem[LDS_F];      // the read addresses XOR bits 4-7: stage bases stay 1 KB-aligned
    __shared__ float dt_s[NSLOT * 64];      // inverse-transform coefficients
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hk = lane >> 5;
    const int team = wave >> 1, par = wave & 1;

    // Persistent workgroups (one per CU: the LDS admits one anyway): workgroup b ranks tiles b, b + gridDim.x, ... Tile numbering
    // in 16 x 16 windows: consecutive tiles walk 16 overhead tiles of one surface tile, then the next surface tile, so the
    // resident workgroups share 16 + 16 tile spectra per slot (L2-resident while the slots advance together).
    const unsigned n_tiles = (unsigned)p.nbx * (unsigned)p.nby;      // < 2^31 (checked by the launcher): 32-bit tile arithmetic
    auto tile_origin = [&](unsigned tile, int& s0_, int& o0_) {
        const unsigned per_group = 16u * (unsigned)p.nbx;
        const int g = (int)(tile / per_group), within = (int)(tile - (unsigned)g * per_group);
        const int rows = min(16, p.nby - 16 * g);
        o0_ = (16 * g + within % rows) * 32;
        s0_ = (within / rows) * 32;
    };

    // ---- staging: LDS-DMA, 16 B per lane: one instruction brings two whole rows (lanes 0-31 row 2n, lanes 32-63 row 2n+1)
    // with no register transit and no ds_write; a lane fetches the 16-byte slot that belongs at its LDS position under the
    // swizzle. Wave w owns the rows of one kind: w&1 = slot parity, w>>1 = 0 surfaces / 1 overheads; 16 instructions per
    // step. Rows past the batch fall outside the descriptor (zeros); every slot a step reads exists (32 slots, 16 steps).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int srp = wv & 1, is_ov = wv >> 1;
    const unsigned region = (is_ov ? A_F + srp * 32 * ROW_F : srp * 32 * ROW_F) * 4u;
    const unsigned lds0 = lds_address(smem);
    // lane -> (row 2n + hi, physical slot l32): logical slot = l32 ^ ((2n + hi) & 15) = (l32 ^ hi) ^ (2n & 15)
    const unsigned voff0 = (unsigned)hk * (SPEC * 4u) + (unsigned)(l31 ^ hk) * 16u;
    auto tile_rsrc = [&](int s0_, int o0_) {
        const int rows_here = is_ov ? min(32, p.Bo - o0_) : min(32, p.Bs - s0_);
        return raw_rsrc(is_ov ? p.spec_ov + (size_t)o0_ * SPEC : p.spec_su + (size_t)s0_ * SPEC, (unsigned)rows_here * SPEC * 4u);
    };
    i32x4 rs;
    auto dma_rows = [&](const i32x4& rs, int n, int step, int buf) {      // n = row pair, compile-time after unrolling
        const unsigned slot = (unsigned)(2 * step + srp);
        const unsigned soff = (unsigned)(2 * n) * (SPEC * 4u) + slot * 512u;
        const unsigned lds = lds0 + (unsigned)buf * (STAGE_F * 4u) + region + (unsigned)n * 1024u;
        dma16(rs, lds, voff0 ^ (unsigned)(((2 * n) & 15) << 4), soff);
    };

    // ---- operand roles. GEMM-1 row l31 = surface j, part (0: Re C, 1: Im C); row order j&3 + 4*part + 8*(j>>2)
    const int j = (l31 & 3) + 4 * (l31 >> 3), part = (l31 >> 2) & 1;
    //   K < 64 (lines x re of the overhead):  Re row reads P, Im row reads Q (sign below);  K >= 64 (x im): Re row reads Q, Im row reads P
    // byte offsets in a stage of k-group 0; k-group u is at offset ^ (u << 4) (the swizzle: slot u ^ (row & 15))
    // 8-byte chunk of a 16-byte slot: ds_read_b64 serves lanes 0-31 and 32-63 in one LDS cycle each when their 32 x 8 bytes fall on
    // 64 distinct banks. A surface's P and Q halves are 256 B apart (the same banks) and are read by the part-0 / part-1 lanes of
    // one instruction, so the surface spectra are STORED with the two chunks of every Q slot exchanged (match_spectrum_kernel,
    // role 0) and a lane reading Q takes chunk hk ^ 1: P readers sit on banks 4c+{0,1}, Q readers on 4c+{2,3}.
    // the odd wave takes the team's surfaces in the order j ^ 8: its accumulator r then belongs to surface r ^ 8, and in the
    // epilogue BOTH waves keep registers 0-7 (the E wave surfaces 0-7, the O wave 8-15) and send registers 8-15
    const int jr = j ^ (par << 3);
    const unsigned a_row = (unsigned)(par * 32 + team * 16 + jr) * 512u + ((unsigned)jr << 4);
    const unsigned a_off1 = a_row + (part ? 256u + 8u * (hk ^ 1) : 8u * hk);
    const unsigned a_off2 = a_row + (part ? 8u * hk : 256u + 8u * (hk ^ 1));
    // slot 0 (the even waves' first step) holds two REAL spectra, P = X_0 and Q = X_32: there the Im rows read what the Re rows
    // read, so that lanes 0-31 of the accumulators end up with C_0 + C_32 and lanes 32-63 (C = cb + sg * ca) with C_32 - C_0 --
    // the coefficient table of slot 0 turns them into (C_0 + (-1)^shift C_32) / 64
    const unsigned a_off1z = par ? a_off1 : a_row + 8u * hk;
    const unsigned a_off2z = par ? a_off2 : a_row + 256u + 8u * (hk ^ 1);
    // the Im rows' minus sign (K < 64: -Q) is applied once per step: ca collects K < 64, cb K >= 64, and accumulator register r
    // holds Re C in lanes 0-31 and Im C in lanes 32-63, so C = cb + sg * ca with sg = -1 in the upper half-wave
    const float sg = hk ? -1.f : 1.f;
    // overheads: one instruction reads one half of 32 different rows; rows r and r + 16 share the slot swizzle, so the spectra of
    // overheads with bit 4 of their index set are stored with the chunks of every slot exchanged (role 1) and read at hk ^ 1
    const unsigned b_off1 = (unsigned)(A_F + (par * 32 + l31) * ROW_F) * 4u + 8u * (hk ^ (l31 >> 4)) + ((unsigned)(l31 & 15) << 4);     // Q: + 256
    for (int t = tid; t < NSLOT * 64; t += 256) dt_s[t] = p.dtab[t];
    // (a global load of the step's coefficient would sit at the end of every step with its whole latency exposed: ~4.6k cycles per step)

    int s0, o0;
    tile_origin(blockIdx.x, s0, o0);
    rs = tile_rsrc(s0, o0);
#pragma unroll
    for (int n = 0; n < 16; ++n) dma_rows(rs, n, 0, 0);        // the first tile's first stage; a later one rides in the last step of the tile before it

    // the wave's 16 x 16 long-lived accumulation registers; zeroed here and again at the end of every epilogue (behind the norm
    // loads of the output phase, whose latency that hides)
    f32x16 acc2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[r][q] = 0.f;
    const int par_u = wv & 1;            // scalar copy of par: the epilogue's two roles are a uniform branch
    float fzero = 0.f;                   // (behind an empty asm: a register, not a folded constant)
    asm volatile("" : "+v"(fzero));

#if WITW_DFT_PHASES
    const unsigned long long ph_k0 = __builtin_amdgcn_s_memrealtime(), ph_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long ph_s[2] = {0, 0}, ph_last = 0;      // [0] barrier -> end of the next GEMM 1 (GEMM 2 + GEMM 1), [1] the wait + barrier
    unsigned long long ph_steps = 0, ph_epi = 0, ph_t0 = 0, ph_t1 = 0, ph_e[6] = {0, 0, 0, 0, 0, 0}, ph_m[6];      // scalar: s_memrealtime sums over this workgroup's tiles
#endif
    int iter = 0;
#pragma clang loop unroll(disable)
    for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++iter) {
    // the NEXT tile's origin and staging descriptor: its first stage is fetched by the staging DMA of this tile's last step (which
    // has no step of its own to fetch for) into stage 0, free by then -- before, a block of 16 DMA instructions per wave in the
    // epilogue (0.6 us per tile: outside the MFMAs' shadow a DMA instruction costs ~85 cycles of issue)
    int s0n = 0, o0n = 0;
    i32x4 rsn = raw_rsrc(p.spec_ov, 0u);      // no next tile: an empty descriptor (zeros)
    if (tile + gridDim.x < n_tiles) {
        tile_origin(tile + gridDim.x, s0n, o0n);
        rsn = tile_rsrc(s0n, o0n);
    }
    const bool rec = REC && p.stamps && blockIdx.x < 4 && iter == 1 && tid == 0;      // a steady-state tile of the first workgroups
    auto stamp = [&](int k) { if (rec) p.stamps[blockIdx.x * 64 + k] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
#if WITW_DFT_PHASES
    ph_t0 = __builtin_amdgcn_s_memrealtime();
    ph_last = ph_t0;
#endif
    // vmcnt(0), said with the builtin: the staging DMA of the first stage has landed, AND the compiler's own counter bookkeeping
    // enters the step loop clean. As asm only, the norm loads of the previous tile's epilogue stayed "pending" for the compiler; in
    // the value-only instantiation their destination registers are the GEMM-1 accumulators, and it protected them with a
    // vmcnt(0) inside the step loop, right behind the first staging DMA of every step: 244 -> 256 ms on configuration 5
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();
    stamp(1);

    // step i: 64 GEMM-1 MFMAs with the 16 DMA instructions of step i+1's rows issued two per MFMA group in the first 8 groups into
    // the other stage (free since the barrier of step i-1), barrier, then the 16 GEMM-2 MFMAs of step i behind the first operand
    // reads and the coefficient read of step i+1: the barrier sits BETWEEN the two GEMMs, so GEMM 2 (registers only) hides the LDS
    // latency of the next step's first reads and the drain of the last GEMM-1 MFMAs overlaps the barrier wait (round 5; before,
    // every step began with a barrier followed by the address arithmetic and the exposed latency of its first reads).
    // Operands as ds_read_b64: a lane holds k = 4u + 2hk and 4u + 2hk + 1 of its row, i.e. MFMA step 2u + e covers k = 4u + e
    // (lanes 0-31) and 4u + 2 + e (lanes 32-63) -- the same K permutation on both operands. Reads run two groups ahead; the
    // compiler would fuse neighbours into ds_read2_b64 (banked like ds_read_b32), hence the asm.
    constexpr int NQ = 3;      // register slots of the operand ring: reads run two k-groups ahead of the MFMAs (three: no gain)
    f32x2 qa1[NQ], qb1[NQ], qa2[NQ], qb2[NQ];
    unsigned xa1, xb1, xa2;
    const unsigned dt0 = lds_address(dt_s) + (unsigned)(par * 64 + lane) * 4u;      // + step * 512
    // read addresses of the 16 k-groups: base ^ (U << 4) (the slot swizzle), 48 registers that live across the step. They are
    // formed between the GEMM-2 MFMAs of the previous step (groups 0 and 1 at the step head): inside the groups the compiler
    // forms each address in the register the read is about to overwrite, which an MFMA in flight still names as its operand,
    // and the step ran 7 % slower (measured, same box).
    unsigned XA[16], XB[16], XC[16];
#define WITW_DFT_FETCH(U)                                              \
        {                                                              \
            qa1[(U) % NQ] = lds_read64<0>(XA[U]);                       \
            qb1[(U) % NQ] = lds_read64<0>(XB[U]);                       \
            qa2[(U) % NQ] = lds_read64<0>(XC[U]);                       \
            qb2[(U) % NQ] = lds_read64<256>(XB[U]);                     \
        }
#define WITW_DFT_ADDR(U)                                               \
        {                                                              \
            XA[U] = lds_xor<((U) << 4)>(xa1);                          \
            XB[U] = lds_xor<((U) << 4)>(xb1);                          \
            XC[U] = lds_xor<((U) << 4)>(xa2);                          \
        }
#define WITW_DFT_STEP_HEAD(STEP, SLOT0)                                \
        {                                                              \
            const unsigned sb = lds0 + (unsigned)((STEP) & 1) * (STAGE_F * 4u) + tile_zero; \
            xa1 = sb + ((SLOT0) ? a_off1z : a_off1); xb1 = sb + b_off1; xa2 = sb + ((SLOT0) ? a_off2z : a_off2); \
            WITW_DFT_ADDR(0)                                           \
            WITW_DFT_ADDR(1)                                           \
            WITW_DFT_FETCH(0)                                          \
            WITW_DFT_FETCH(1)                                          \
        }
    // tile_zero = 0 behind an empty asm: the 48 step-0 addresses below are the same for every tile, and the compiler would hoist
    // them out of the tile loop and keep them across the epilogue (42 registers; the GAP instantiation then spilled 59, reloaded
    // them behind the tile-top barrier and waited for the reloads -- i.e. for the staging DMA -- inside the step loop)
    unsigned tile_zero = 0;
    asm volatile("" : "+v"(tile_zero));
    WITW_DFT_STEP_HEAD(0, true)
#pragma unroll
    for (int u = 2; u < 16; ++u) {      // (again for every tile: 42 instructions, and the 48 registers are free during the epilogue)
        XA[u] = xa1 ^ (unsigned)(u << 4);
        XB[u] = xb1 ^ (unsigned)(u << 4);
        XC[u] = xa2 ^ (unsigned)(u << 4);
    }
#pragma clang loop unroll(disable)
    for (int i = 0; i < NSTEP; ++i) {
        const int bufn = (i + 1) & 1;
        const bool last_step = i + 1 == NSTEP;
        const int inext = last_step ? 0 : i + 1;      // the last step stages step 0 of the next tile
        i32x4 rsd;
#pragma unroll
        for (int e = 0; e < 4; ++e) rsd[e] = last_step ? rsn[e] : rs[e];
        f32x16 ca, cb;      // the first MFMA of each chain starts from C = 0
        float dval;
#define WITW_DFT_GROUP(U)                                                                                                      \
        {                                                                                                                      \
            constexpr int d = (U) % NQ;                                                                                        \
            constexpr int AH = 2;                                                                                              \
            if ((U) + AH < 16 && !(WITW_DFT_DIAG & 4)) WITW_DFT_FETCH((U) + AH < 16 ? (U) + AH : 0)                            \
            if ((U) + AH < 16) lds_wait<4 * AH>(qa1[d], qb1[d], qa2[d], qb2[d]);                                               \
            else if ((U) + 1 < 16) lds_wait<4 * (15 - (U) < AH ? 15 - (U) : AH)>(qa1[d], qb1[d], qa2[d], qb2[d]);              \
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qa1[d]), "+v"(qb1[d]), "+v"(qa2[d]), "+v"(qb2[d]), "+v"(dval));     \
            if ((U) == 14) dval = lds_read32(dt0 + (unsigned)i * 512u);      /* the step's coefficient: waited for by group 15 */ \
            if ((U) == 0) mfma_v0(ca, qa1[d][0], qb1[d][0]); else mfma_v(ca, qa1[d][0], qb1[d][0]);                            \
            if (!(WITW_DFT_DIAG & 1) && (U) < 8) dma_rows(rsd, 2 * (U), inext, bufn);                                               \
            if ((U) == 0) mfma_v0(cb, qa2[d][0], qb2[d][0]); else mfma_v(cb, qa2[d][0], qb2[d][0]);                            \
            if (!(WITW_DFT_DIAG & 1) && (U) < 8) dma_rows(rsd, 2 * (U) + 1, inext, bufn);                                           \
            mfma_v(ca, qa1[d][1], qb1[d][1]);                                                                                  \
            mfma_v(cb, qa2[d][1], qb2[d][1]);                                                                                  \
        }
        WITW_DFT_GROUP(0) WITW_DFT_GROUP(1) WITW_DFT_GROUP(2) WITW_DFT_GROUP(3)
        WITW_DFT_GROUP(4) WITW_DFT_GROUP(5) WITW_DFT_GROUP(6) WITW_DFT_GROUP(7)
        WITW_DFT_GROUP(8) WITW_DFT_GROUP(9) WITW_DFT_GROUP(10) WITW_DFT_GROUP(11)
        WITW_DFT_GROUP(12) WITW_DFT_GROUP(13) WITW_DFT_GROUP(14) WITW_DFT_GROUP(15)
#undef WITW_DFT_GROUP
        stamp(2 + 3 * i);
#if WITW_DFT_PHASES
        const unsigned long long pa = __builtin_amdgcn_s_memrealtime();
#endif
        if (!(WITW_DFT_DIAG & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#if WITW_DFT_PHASES
        const unsigned long long pb = __builtin_amdgcn_s_memrealtime();
        ph_s[0] += pa - ph_last; ph_s[1] += pb - pa; ph_last = pb;
#endif
        stamp(3 + 3 * i);
        if (i + 1 < NSTEP) WITW_DFT_STEP_HEAD(i + 1, false)
        mfma_settle(ca, cb);
        // C = cb + sg * ca into 16 DIFFERENT registers before the first GEMM-2 MFMA: left to the compiler every product went through
        // one register, and a VALU write to a register that the MFMA in flight names as its operand waits for that MFMA -- each
        // of the 16 GEMM-2 MFMAs then cost ~90 cycles instead of 64 (the same effect as the address registers above)
        float cc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cc[r] = fmaf(ca[r], sg, cb[r]);
        asm volatile("" : "+v"(cc[0]), "+v"(cc[1]), "+v"(cc[2]), "+v"(cc[3]), "+v"(cc[4]), "+v"(cc[5]), "+v"(cc[6]), "+v"(cc[7]),
                          "+v"(cc[8]), "+v"(cc[9]), "+v"(cc[10]), "+v"(cc[11]), "+v"(cc[12]), "+v"(cc[13]), "+v"(cc[14]), "+v"(cc[15]));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc2[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(dval, cc[r], acc2[r], 0, 0, 0);      // [shift][overhead]
            if (r >= 2) {      // the next step's read addresses, in the shadow of this MFMA (groups 0 and 1: at the step head)
                XA[r] = xa1 ^ (unsigned)(r << 4);
                XB[r] = xb1 ^ (unsigned)(r << 4);
                XC[r] = xa2 ^ (unsigned)(r << 4);
            }
        }
        // the 16 operand registers stay live up to here: otherwise the address registers above are allocated on top of them. The
        // last accumulator is named as well: it ties this statement behind the last MFMA (an empty asm may move above the builtins)
        asm volatile("" : "+a"(acc2[15]) : "v"(cc[0]), "v"(cc[1]), "v"(cc[2]), "v"(cc[3]), "v"(cc[4]), "v"(cc[5]), "v"(cc[6]), "v"(cc[7]),
                           "v"(cc[8]), "v"(cc[9]), "v"(cc[10]), "v"(cc[11]), "v"(cc[12]), "v"(cc[13]), "v"(cc[14]), "v"(cc[15]), "v"(dval));
        stamp(4 + 3 * i);
    }
#undef WITW_DFT_STEP_HEAD
#undef WITW_DFT_FETCH
#undef WITW_DFT_ADDR

#if WITW_DFT_PHASES
    ph_t1 = __builtin_amdgcn_s_memrealtime();
    ph_steps += ph_t1 - ph_t0;
#endif
    // ---- the epilogue runs in the area of stage 1; stage 0 already holds the next tile's first stage (fetched by the last step)
    const int s0c = s0, o0c = o0;
    s0 = s0n; o0 = o0n; rs = rsn;      // the tile whose first stage is on its way
    if (WITW_DFT_DIAG & 8) {
        float t = 0.f;      // every accumulator stays live
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int q = 0; q < 16; ++q) t += acc2[r][q];
        if (t == 12345.f && p.score) p.score[tile] = t;
        continue;
    }
    // ---- epilogue. acc2[r][q] of lane (l31, hk) = E (even wave) or O (odd wave) of surface r of the team, overhead l31, shift
    // (q & 3) + 8 (q >> 2) + 4 hk (the odd wave: surface r ^ 8). Two rounds h: a wave sends registers 8 + 4h .. +3 (the partner's
    // surfaces) and receives the partner's tiles of its own surfaces, registers 4h .. +3; LDS [wave][surface of the round][register quad][lane] x 16 B (conflict-free both ways).
    float* xw = smem + STAGE_F + wv * 4096 + lane * 4;
    const float* xr = smem + STAGE_F + (wv ^ 1) * 4096 + lane * 4;
    float rv[8], rs[8];
    int rk[8];
    // value-only: the norms of the output phase depend on no result (window norm of shift 0) -- loaded here, a whole epilogue ahead
    float wn_e = 1.f, sn_e[4] = {1.f, 1.f, 1.f, 1.f};
    if (VONLY && p.distance) {
        const int og = o0c + l31;
        if (og < p.Bo) wn_e = p.wn[(size_t)og * 64];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int s = s0c + team * 16 + par * 8 + 4 * hk + jj;
            if (s < p.Bs) sn_e[jj] = p.sn[s];
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const f32x4 t = {acc2[8 + 4 * h + rr][4 * qq], acc2[8 + 4 * h + rr][4 * qq + 1], acc2[8 + 4 * h + rr][4 * qq + 2], acc2[8 + 4 * h + rr][4 * qq + 3]};
                *reinterpret_cast<f32x4*>(xw + (rr * 4 + qq) * 256) = t;
            }
        // the sent registers are zeroed for the next tile here, in front of the barrier wait
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) mfma_zero(acc2[8 + 4 * h + rr], fzero);
        __syncthreads();
#if WITW_DFT_PHASES
        asm volatile("" ::: "memory");
        ph_m[2 * h] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            f32x16 got;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(xr + (rr * 4 + qq) * 256);
                got[4 * qq] = t[0]; got[4 * qq + 1] = t[1]; got[4 * qq + 2] = t[2]; got[4 * qq + 3] = t[3];
            }
            float v, s2 = 0.f;
            unsigned k;
            if (par_u == 0) scan16<GAP, VONLY>(acc2[4 * h + rr], got, v, k, s2);      // kept: E, received: O
            else scan16<GAP, VONLY>(got, acc2[4 * h + rr], v, k, s2);
            // the other 16 shifts of the pair sit in lane ^ 32 (4 further on for the upper half-wave)
            float v0, v1;
            unsigned kl, ku0;
            half_swap(v, v0, v1);
            const float vc = fmaxf(v0, v1);
            int kc = 0;
            if (!VONLY) {
                half_swap(k, kl, ku0);
                const unsigned ku = ku0 == NOKEY ? NOKEY : ku0 + 4u;
                const unsigned kb = umin(v0 == vc ? kl : NOKEY, v1 == vc ? ku : NOKEY);
                kc = kb == NOKEY ? 0 : (int)((kb & 63u) + ((kb >> 31) << 5));      // no finite maximum (NaN scores): index 0
            }
            rv[4 * h + rr] = vc;
            rk[4 * h + rr] = kc;
            if (GAP) {
                float s0, s1;
                half_swap(s2, s0, s1);
                rs[4 * h + rr] = fmaxf(fmaxf(s0, s1), fminf(v0, v1));
            }
        }
#if WITW_DFT_PHASES
        asm volatile("" :: "v"(rv[4 * h]), "v"(rv[4 * h + 1]), "v"(rv[4 * h + 2]), "v"(rv[4 * h + 3]), "v"(rk[4 * h + 3]) : "memory");
        ph_m[2 * h + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        // this round's kept accumulators are dead: back to zero on the matrix pipe while the vector pipe goes on
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) mfma_zero(acc2[4 * h + rr], fzero);
        if (GAP) {
            // this instantiation sits at the register limit: a round's four results are written at once (two surfaces per
            // half-wave) instead of being carried to the end of the epilogue
            const int og = o0c + l31;
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const float v = hk ? rv[4 * h + 2 + j2] : rv[4 * h + j2];
                const int kx = hk ? rk[4 * h + 2 + j2] : rk[4 * h + j2];
                const float g = hk ? rs[4 * h + 2 + j2] : rs[4 * h + j2];
                const int s = s0c + team * 16 + par * 8 + 4 * h + 2 * hk + j2;
                if (s < p.Bs && og < p.Bo) {
                    const size_t off = (size_t)og * p.Bs + s;
                    if (p.orientation) p.orientation[off] = kx;
                    if (p.score) p.score[off] = v;
                    if (p.distance) p.distance[off] = 2.f * (1.f - v / (p.wn[(size_t)og * 64 + kx] * p.sn[s]));
                    if (p.gap) p.gap[off] = v - g;
                }
            }
        }
        if (h == 0) __syncthreads();      // the partner has read round 0 before round 1 overwrites it
    }
    // ---- output: both half-waves hold the 8 results of overhead l31; the lower one writes surfaces 0-3, the upper one 4-7. The
    // window-norm loads are issued first, the accumulators are zeroed for the next tile behind them
    if (!GAP) {
        const int og = o0c + l31;
        float wnv[4], snv[4], vv[4];
        int kk[4];
        bool ok[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            vv[jj] = hk ? rv[4 + jj] : rv[jj];
            kk[jj] = hk ? rk[4 + jj] : rk[jj];
            const int s = s0c + team * 16 + par * 8 + 4 * hk + jj;
            ok[jj] = s < p.Bs && og < p.Bo;
            wnv[jj] = VONLY ? wn_e : (ok[jj] && p.distance) ? p.wn[(size_t)og * 64 + kk[jj]] : 1.f;
            snv[jj] = VONLY ? sn_e[jj] : (ok[jj] && p.distance) ? p.sn[s] : 1.f;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int s = s0c + team * 16 + par * 8 + 4 * hk + jj;
            if (ok[jj]) {
                const size_t off = (size_t)og * p.Bs + s;
                if (p.orientation) p.orientation[off] = kk[jj];
                if (p.score) p.score[off] = vv[jj];
                if (p.distance) p.distance[off] = 2.f * (1.f - vv[jj] / (wnv[jj] * snv[jj]));
            }
        }
    }
    stamp(53);
#if WITW_DFT_PHASES
    asm volatile("" ::: "memory");
    {
        const unsigned long long te = __builtin_amdgcn_s_memrealtime();
        ph_epi += te - ph_t1;
        ph_e[0] += ph_m[0] - ph_t1; ph_e[1] += ph_m[1] - ph_m[0]; ph_e[2] += ph_m[2] - ph_m[1]; ph_e[3] += ph_m[3] - ph_m[2]; ph_e[4] += te - ph_m[3];
    }
#endif
    }   // tiles
#if WITW_DFT_PHASES
    if (p.stamps && blockIdx.x < 4 && tid == 0) {
        p.stamps[blockIdx.x * 64 + 0] = ph_steps;
        p.stamps[blockIdx.x * 64 + 1] = ph_epi;
        p.stamps[blockIdx.x * 64 + 2] = (unsigned long long)iter;
        for (int e = 0; e < 5; ++e) p.stamps[blockIdx.x * 64 + 3 + e] = ph_e[e];
        p.stamps[blockIdx.x * 64 + 10] = __builtin_amdgcn_s_memrealtime() - ph_k0;
        p.stamps[blockIdx.x * 64 + 11] = __builtin_amdgcn_s_memtime() - ph_c0;
        p.stamps[blockIdx.x * 64 + 8] = ph_s[0];
        p.stamps[blockIdx.x * 64 + 9] = ph_s[1];
    }
#endif
}

// spec[e][t][0..63] = Re X_t(line), [64..127] = Im X_t(line) (0 for t = 0, 32), X_t = sum_k x[line][k] e^{-2 pi i t k / 64}; fp64
// accumulation, rounded once to fp32. One workgroup per embedding [64 lines][W columns], W <= 64.
// role: whose spectrum this is, which fixes the order of the two 8-byte chunks (float pairs) inside each 16-byte slot so that the
// match kernel's ds_read_b64 are free of bank conflicts (see the operand roles there): 0 = surface (queries): the chunks of the
// Q half exchanged; 1 = overhead (gallery): both halves' chunks exchanged for embeddings whose index has bit 4 set.
__global__ __launch_bounds__(256) void match_spectrum_kernel(const float* __restrict__ emb, float* __restrict__ spec, int W, int role) {
    __shared__ float xs[64 * 65];
    __shared__ double cs[64], sn[64];
    const int tid = threadIdx.x, line = tid & 63, tq = tid >> 6;
    const float* x = emb + (size_t)blockIdx.x * 64 * W;
    for (int i = tid; i < 64 * W; i += 256) xs[(i / W) * 65 + (i % W)] = x[i];
    if (tid < 64) {
        cs[tid] = cospi((double)tid / 32.0);
        sn[tid] = sinpi((double)tid / 32.0);
    }
    __syncthreads();
    float* out = spec + (size_t)blockIdx.x * SPEC;
    for (int t = tq; t <= 32; t += 4) {      // frequencies 0..32; X_0 and X_32 are real and share storage slot 0 (P = X_0, Q = X_32)
        double pr = 0.0, pi = 0.0;
        for (int k = 0; k < W; ++k) {
            const int idx = (t * k) & 63;
            const double xv = (double)xs[line * 65 + k];
            pr += xv * cs[idx];
            pi -= xv * sn[idx];
        }
        const int swap_p = (role == 1 && (blockIdx.x & 16)) ? 2 : 0;      // float index ^ 2 = the other 8-byte chunk of the slot
        const int swap_q = (role == 0 || (blockIdx.x & 16)) ? 2 : 0;
        if (t == 32) out[64 + (line ^ swap_q)] = (float)pr;
        else {
            out[t * 128 + (line ^ swap_p)] = (float)pr;
            if (t != 0) out[t * 128 + 64 + (line ^ swap_q)] = (float)pi;
        }
    }
}

// The same spectra for full-width embeddings (W = 64: every gallery row, and the queries at fov 360) with the inner loop in
// registers (round 6). The kernel above reads three LDS words per (line, frequency, k) -- the sample, cos and sin as doubles -- and
// is bound by those reads: 125,000 gallery rows took 5 ms of the retrieval pass. Here a thread reads its line's samples (once per k from LDS), the 17 distinct twiddle magnitudes cos(2 pi m / 64), m = 0..16, are registers too, and because
// a wave's frequencies are fixed (wave 0 the even t, wave 1 the odd t) every (t, k) names its magnitude and sign when the body is
// generated: two v_fma_f64 per (line, t, k), one LDS word and one convert per (line, k). Same sums in the same k order, fp64, rounded once to fp32; products are fused here
// (the kernel above rounds the product, then the sum), so single values may differ in the last fp32 bit -- inside the rounding
// bound of the spectral scores that the index-exact re-scoring is built on (ops.SCORE_ROUNDING).
struct Twiddle64 { double c[17]; };      // cos(2 pi m / 64), m = 0..16: kernel arguments, i.e. scalar registers

// one workgroup = two waves per embedding: lane = line, wave = parity of the frequencies it sums; the body is generated
// (tools/gen_spectrum64.py -> spectrum64_gen.h): straight-line code, every twiddle a named scalar with its sign
__global__ __launch_bounds__(128) void match_spectrum64_kernel(const float* __restrict__ emb, float* __restrict__ spec, int role, Twiddle64 tw) {
    __shared__ float xs[64 * 65];
    const int tid = threadIdx.x, line = tid & 63;
    const int par = __builtin_amdgcn_readfirstlane(tid >> 6);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(emb + (size_t)blockIdx.x * 4096);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + i * 128;              // float4 index: row q >> 4, columns 4 (q & 15) ..
        const f32x4 v = x4[q];
        float* d = xs + (q >> 4) * 65 + 4 * (q & 15);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    const double c0 = tw.c[0], c1 = tw.c[1], c2 = tw.c[2], c3 = tw.c[3], c4 = tw.c[4], c5 = tw.c[5], c6 = tw.c[6], c7 = tw.c[7], c8 = tw.c[8],
                 c9 = tw.c[9], c10 = tw.c[10], c11 = tw.c[11], c12 = tw.c[12], c13 = tw.c[13], c14 = tw.c[14], c15 = tw.c[15], c16 = tw.c[16];
    __syncthreads();
    float* out = spec + (size_t)blockIdx.x * SPEC;
    const int swap_p = (role == 1 && (blockIdx.x & 16)) ? 2 : 0;      // as in match_spectrum_kernel
    const int swap_q = (role == 0 || (blockIdx.x & 16)) ? 2 : 0;
    const float* xrow = xs + line * 65;           // pitch 65: the wave's 64 lines read 64 different banks
#define WITW_SPEC_STORE(T, PR, PI)                                            \
    if ((T) == 32) out[64 + (line ^ swap_q)] = (float)(PR);                   \
    else {                                                                    \
        out[(T) * 128 + (line ^ swap_p)] = (float)(PR);                       \
        if ((T) != 0) out[(T) * 128 + 64 + (line ^ swap_q)] = (float)(PI);    \
    }
    if (par == 0) {
        WITW_SPECTRUM64_BODY_0(xrow);
        WITW_SPECTRUM64_STORE_0(WITW_SPEC_STORE);
    } else {
        WITW_SPECTRUM64_BODY_1(xrow);
        WITW_SPECTRUM64_STORE_1(WITW_SPEC_STORE);
    }
#undef WITW_SPEC_STORE
}

// dtab[t][hk*32 + shift]: coefficient of lanes 0-31 (hk = 0) / 32-63 (hk = 1) of slot t's C in score[shift], shift < 32. Slots
// 1..31: Re C_t / Im C_t. Slot 0: the kernel leaves C_0 + C_32 in lanes 0-31 and C_32 - C_0 in lanes 32-63, and
// (C_0 + (-1)^shift C_32) / 64 is the former at even shifts and minus the latter at odd ones
__global__ void match_dft_table_kernel(float* __restrict__ dtab) {
    const int t = blockIdx.x, lane = threadIdx.x, shift = lane & 31, hk = lane >> 5;
    double v;
    if (t == 0) v = hk ? ((shift & 1) ? -1.0 / 64.0 : 0.0) : ((shift & 1) ? 0.0 : 1.0 / 64.0);
    else {
        const double ang = (double)((t * shift) & 63) / 32.0;
        v = hk ? -sinpi(ang) / 32.0 : cospi(ang) / 32.0;
    }
    dtab[t * 64 + lane] = (float)v;
}

// the two norm kernels of match.hip's launch, restated here (file-local there)
__global__ __launch_bounds__(256) void dft_window_norm_kernel(const float* __restrict__ ov, float* __restrict__ wn, int We) {
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

__global__ __launch_bounds__(256) void dft_row_norm_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
    __shared__ float part[4];
    const float* base = x + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = base[i];
        s += v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sqrtf((part[0] + part[1]) + (part[2] + part[3]));
}

}  // namespace

extern "C" {

// floats of one embedding's row spectrum (32 slots x [64 re | 64 im]; slot 0: [X_0 | X_32])
long long witw_match_spectrum_floats(long long n_embeddings) { return n_embeddings * (long long)SPEC; }

// emb [B,64 lines,W] (an overhead embedding [B,16,4,64], role 1, or a surface embedding [B,16,4,We], role 0) -> spec [B,32,128]
// in the chunk order the match kernel's operand reads expect of that side (an overhead's spectrum depends on its index & 16:
// spectra of a gallery must be computed at the row numbering they are matched at, multiples of 32 apart)
static int g_spectrum_regs = getenv("WITW_SPECTRUM_LDS") == nullptr;      // WITW_SPECTRUM_LDS=1: the LDS-table kernel at every width (A/B)

int witw_match_spectrum(const float* emb, float* spec, int B, int W, int role, void* stream) {
    WITW_CHECK_ARG(emb && spec, "match_spectrum: null pointer");
    WITW_CHECK_ARG(B > 0 && W >= 1 && W <= 64, "match_spectrum: bad shape B=%d W=%d", B, W);
    WITW_CHECK_ARG(role == 0 || role == 1, "match_spectrum: role %d (0 = surface / query side, 1 = overhead / gallery side)", role);
    if (W == 64 && g_spectrum_regs)
        {
        Twiddle64 tw;
        for (int m = 0; m <= 16; ++m) tw.c[m] = m == 16 ? 0.0 : cos(2.0 * 3.14159265358979323846 * m / 64.0);
        hipLaunchKernelGGL(match_spectrum64_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, emb, spec, role, tw);
    }
    else
        hipLaunchKernelGGL(match_spectrum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, emb, spec, W, role);
    WITW_CHECK_LAUNCH("match_spectrum");
    return WITW_OK;
}

long long witw_match_dft_workspace_floats(int Bo, int Bs) { return (long long)Bo * 64 + Bs + NSLOT * 64; }

// Same outputs as witw_match_fwd (orientation / distance / score [Bo,Bs], any of them may be null) from the row spectra of the
// two sides (witw_match_spectrum of ov with W = 64 and of su with W = We); ov / su themselves are read for the norms only.
static int match_fwd_dft_launch(const float* ov, const float* su, const float* spec_ov, const float* spec_su, int Bo, int Bs, int We,
                                long long* orientation, float* distance, float* score, float* gap, float* workspace, void* stream) {
    WITW_CHECK_ARG(ov && su && spec_ov && spec_su && workspace, "match_fwd_dft: null pointer");
    WITW_CHECK_ARG(Bo > 0 && Bs > 0, "match_fwd_dft: empty batch Bo=%d Bs=%d", Bo, Bs);
    WITW_CHECK_ARG(We >= 1 && We <= 64, "match_fwd_dft: surface embedding width %d outside [1,64]", We);
    hipStream_t st = (hipStream_t)stream;
    float* wn = workspace;
    float* sn = workspace + (size_t)Bo * 64;
    float* dtab = sn + Bs;
    hipLaunchKernelGGL(dft_window_norm_kernel, dim3(Bo), dim3(256), 0, st, ov, wn, We);
    hipLaunchKernelGGL(dft_row_norm_kernel, dim3(Bs), dim3(256), 0, st, su, sn, 64 * We);
    hipLaunchKernelGGL(match_dft_table_kernel, dim3(NSLOT), dim3(64), 0, st, dtab);
    DftArgs a;
    a.spec_ov = spec_ov; a.spec_su = spec_su; a.dtab = dtab; a.wn = wn; a.sn = sn;
    a.orientation = orientation; a.distance = distance; a.score = score; a.gap = gap;
    a.Bo = Bo; a.Bs = Bs; a.nbx = cdiv(Bs, 32); a.nby = cdiv(Bo, 32);
    const long long tiles = (long long)a.nbx * a.nby;
    WITW_CHECK_ARG(tiles < (1LL << 31), "match_fwd_dft: %lld tiles of 32 x 32 pairs (the kernel counts tiles in 32 bits)", tiles);
    const int n_cu = witw_cu_count();        // persistent workgroups, one per CU
    const unsigned grid = (unsigned)(tiles < n_cu ? tiles : n_cu);
    // WITW_DFT_STAMPS=1 (diagnostic, synchronous): the first workgroups record s_memrealtime around the phases of their second
    // tile; printed to stderr
    a.stamps = nullptr;
    const int nrec = 4;
    if (!gap && getenv("WITW_DFT_STAMPS") != nullptr && tiles >= 2LL * grid) {
        if (hipMalloc((void**)&a.stamps, (size_t)nrec * 64 * 8) != hipSuccess) a.stamps = nullptr;
        else (void)hipMemset(a.stamps, 0, (size_t)nrec * 64 * 8);
    }
    // value-only scan: no orientation wanted and the surface as wide as the overhead (fov 360: the window norm is the same sum for
    // every shift, in another order -- the distance then uses shift 0's, within an ulp of any other's)
    const bool vonly = !orientation && !gap && We == 64 && getenv("WITW_DFT_VONLY_OFF") == nullptr;
    if (gap) hipLaunchKernelGGL((match_dft_kernel<false, true, false>), dim3(grid), dim3(256), 0, st, a);
    else if (a.stamps && WITW_DFT_PHASES && getenv("WITW_DFT_STAMPS")[0] == '2') {
        if (vonly) hipLaunchKernelGGL((match_dft_kernel<false, false, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((match_dft_kernel<false, false, false>), dim3(grid), dim3(256), 0, st, a);
        (void)hipDeviceSynchronize();
        unsigned long long h[4 * 64];
        (void)hipMemcpy(h, a.stamps, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; ++b)
            fprintf(stderr, "match_dft workgroup %d: %llu tiles, steps %.2f us per tile, epilogue %.2f us per tile (send 0 + barrier %.2f, scans 0 %.2f, "
                    "barrier + send 1 + barrier %.2f, scans 1 %.2f, output + zeroing %.2f)\n", b, h[b * 64 + 2],
                    h[b * 64] * 0.01 / (double)h[b * 64 + 2], h[b * 64 + 1] * 0.01 / (double)h[b * 64 + 2], h[b * 64 + 3] * 0.01 / (double)h[b * 64 + 2],
                    h[b * 64 + 4] * 0.01 / (double)h[b * 64 + 2], h[b * 64 + 5] * 0.01 / (double)h[b * 64 + 2], h[b * 64 + 6] * 0.01 / (double)h[b * 64 + 2],
                    h[b * 64 + 7] * 0.01 / (double)h[b * 64 + 2]);
        for (int b = 0; b < 4; ++b)
            fprintf(stderr, "match_dft workgroup %d: per step, previous barrier -> end of GEMM 1 %.3f us, vmcnt wait + barrier %.3f us; s_memtime ticks per us %.1f\n", b,
                    h[b * 64 + 8] * 0.01 / ((double)NSTEP * (double)h[b * 64 + 2]), h[b * 64 + 9] * 0.01 / ((double)NSTEP * (double)h[b * 64 + 2]),
                    (double)h[b * 64 + 11] / ((double)h[b * 64 + 10] * 0.01));
        (void)hipFree(a.stamps);
        a.stamps = nullptr;
    }
    else if (a.stamps) hipLaunchKernelGGL((match_dft_kernel<true, false, false>), dim3(grid), dim3(256), 0, st, a);
    else if (vonly) hipLaunchKernelGGL((match_dft_kernel<false, false, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((match_dft_kernel<false, false, false>), dim3(grid), dim3(256), 0, st, a);
    if (a.stamps) {
        (void)hipDeviceSynchronize();
        unsigned long long* h = (unsigned long long*)malloc((size_t)nrec * 64 * 8);
        (void)hipMemcpy(h, a.stamps, (size_t)nrec * 64 * 8, hipMemcpyDeviceToHost);
        for (int b = 0; b < nrec && b < 4; ++b) {
            const unsigned long long* t = h + (size_t)b * 64;
            fprintf(stderr, "match_dft workgroup %d, second tile: first-stage wait %.2f us; steps (gemm1, barrier, gemm2) us:", b, (t[1] - t[0]) * 0.01);
            for (int i = 0; i < NSTEP; ++i)
                fprintf(stderr, " [%.2f %.2f %.2f]", (t[2 + 3 * i] - (i ? t[1 + 3 * i] : t[1])) * 0.01, (t[3 + 3 * i] - t[2 + 3 * i]) * 0.01,
                        (t[4 + 3 * i] - t[3 + 3 * i]) * 0.01);
            fprintf(stderr, "; epilogue %.2f us; total %.2f us\n", (t[53] - t[1 + 3 * NSTEP]) * 0.01, (t[53] - t[0]) * 0.01);
        }
        free(h);
        (void)hipFree(a.stamps);
    }
    WITW_CHECK_LAUNCH("match_fwd_dft");
    return WITW_OK;
}

int witw_match_fwd_dft(const float* ov, const float* su, const float* spec_ov, const float* spec_su, int Bo, int Bs, int We,
                       long long* orientation, float* distance, float* score, float* workspace, void* stream) {
    return match_fwd_dft_launch(ov, su, spec_ov, spec_su, Bo, Bs, We, orientation, distance, score, nullptr, workspace, stream);
}

// witw_match_fwd_dft that also writes gap [Bo,Bs] = best score - runner-up score of every pair (how far the chosen shift is
// from a tie): the caller re-scores the pairs whose gap is within rounding with witw_match_pairs.
int witw_match_fwd_dft_gap(const float* ov, const float* su, const float* spec_ov, const float* spec_su, int Bo, int Bs, int We,
                           long long* orientation, float* distance, float* score, float* gap, float* workspace, void* stream) {
    WITW_CHECK_ARG(gap, "match_fwd_dft_gap: null gap pointer");
    return match_fwd_dft_launch(ov, su, spec_ov, spec_su, Bo, Bs, We, orientation, distance, score, gap, workspace, stream);
}

}  // extern "C"
