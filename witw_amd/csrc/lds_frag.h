// Hand-issued LDS fragment reads with counted waits (gfx950): shared by the weight-resident bf16 kernels
// (conv_first2_bf16.hip, conv3x3_bf16_wres.hip).
//
// Left to the scheduler each ds_read sinks to just in front of its MFMA and every MFMA waits on lgkmcnt(0). asm volatile
// statements keep their order and LDS returns in order, so a consumer can wait with lgkmcnt(number of reads issued after the last
// one it needs); lds_wait names the fragments it releases, so that the MFMAs reading them cannot be scheduled above it.
#pragma once
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
typedef unsigned int u32x2_frag __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2_frag lds_read64(unsigned addr) {
    u32x2_frag v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// s_waitcnt lgkmcnt(0) over a whole array of fragments (everything this wave has in the LDS queue has landed)
template <typename T, int N>
__device__ __forceinline__ void lds_wait_all(T (&f)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));      // the consumers of f[i] stay behind the wait
}

// s_waitcnt lgkmcnt(n) that names the fragments it releases, so that the MFMAs reading them cannot be scheduled above it
__device__ __forceinline__ void lds_wait(int n, u32x4& a) {
    switch (n) {      // n is a constant once the caller's loop is unrolled
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a)); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a)); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a)); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a)); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a)); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a)); break;
    case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a)); break;
    case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a)); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(a)); break;
    case 10: asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(a)); break;
    case 11: asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(a)); break;
    case 12: asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(a)); break;
    case 13: asm volatile("s_waitcnt lgkmcnt(13)" : "+v"(a)); break;
    case 14: asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(a)); break;
    default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a)); break;      // the counter field has 4 bits: deeper queues wait here (conservative)
    }
}
__device__ __forceinline__ void lds_wait(int n, u32x4& a, u32x4& b) {
    switch (n) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a), "+v"(b)); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b)); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a), "+v"(b)); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b)); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a), "+v"(b)); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a), "+v"(b)); break;
    case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a), "+v"(b)); break;
    case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(b)); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(a), "+v"(b)); break;
    case 10: asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(a), "+v"(b)); break;
    case 11: asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(a), "+v"(b)); break;
    case 12: asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(a), "+v"(b)); break;
    case 13: asm volatile("s_waitcnt lgkmcnt(13)" : "+v"(a), "+v"(b)); break;
    case 14: asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(a), "+v"(b)); break;
    default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a), "+v"(b)); break;      // the counter field has 4 bits: deeper queues wait here (conservative)
    }
}

}  // namespace
