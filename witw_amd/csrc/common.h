// Shared helpers for the WITW gfx950 kernels (HIP, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Error convention of the C ABI (include/witw_hip.h): 0 = ok, negative = error,
// message retrievable through witw_last_error() (thread-local).
enum {
    WITW_OK = 0,
    WITW_ERR_INVALID = -1,   // bad argument / unsupported shape
    WITW_ERR_LAUNCH = -2,    // HIP launch failure
    WITW_ERR_NODEVICE = -3,  // no gfx950 device
};

void witw_set_error(const char* fmt, ...);
void witw_note_variant(const char* fmt, ...);   // records the launched instantiation (witw_last_kernel_variant)

#define WITW_CHECK_ARG(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            witw_set_error(__VA_ARGS__);         \
            return WITW_ERR_INVALID;             \
        }                                        \
    } while (0)

#define WITW_CHECK_LAUNCH(what)                                            \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) {                                            \
            witw_set_error("%s: %s", what, hipGetErrorString(e_));         \
            return WITW_ERR_LAUNCH;                                        \
        }                                                                  \
    } while (0)

int witw_cu_count();                        // api.hip
bool witw_fills_rounds(long long workgroups);

// conv3x3_bf16_wres.hip: the weight-resident 64-input-channel bf16 forward, chosen by witw_conv3x3_bf16_fwd_ex
bool witw_bf16_wres_applies(int B, int H, int W, int Cin, int Cout);
int witw_bf16_wres_launch(const void* x, const void* wpk, const float* bias, const void* gate, const void* gate_bits, void* y, int B, int H,
                          int W, int Cout, int pad_circular, int relu, void* stream);      // gate: null, or the dgrad launch's ReLU gate (a tensor shaped like y)

// ReLU on PACKED bf16 pairs: max as signed 16-bit integers against `floor2` -- a negative float is a negative integer (and -0
// becomes +0), a positive one is itself; floor2 = 0 is the ReLU, 0x80008000 (the smallest integers) leaves the pair unchanged, so
// a run-time "relu" flag costs no select. One v_pk_max_i16 per two values, after the conversion, instead of a v_max_f32 (two when
// the operand has to be canonicalised first) and a select per value. Same bits as relu-then-round (rounding keeps the sign);
// a NaN with the sign bit clear stays a NaN.
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned witw_pack_bf16x2(float lo, float hi) {
    // the instruction (__bf16)x compiles to (round to nearest even), written out: left to the compiler a pair that is then used as
    // an integer is converted one value at a time and joined by a v_perm_b32
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned witw_relu_bf16x2(unsigned pair, unsigned floor2) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pair), __builtin_bit_cast(s16x2, floor2)));
}
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
