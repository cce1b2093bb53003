// Helpers shared by the VALU-free tiled kernels (qs_gemm_fast.hip, qs_gemm_strip.hip): XCD-chunked work order, wave-uniform
// values in scalar registers, global items through SGPR buffer descriptors (scalar base + 32-bit lane offset + range check).
#pragma once

#include "qs_common.h"

namespace qs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Work item of virtual block `v` of a `total`-block grid: XCD x (= v % 8) owns
// the x-th contiguous chunk of the work list; bijective for every `total`.
__device__ __forceinline__ unsigned xcd_chunked_index_fast(unsigned v, unsigned total) {
    const unsigned xcd = v & 7u, slot = v >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// value known to be wave-uniform -> scalar registers
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// One global item (16 or 8 bytes) at scalar base + 32-bit lane offset, buffer form:
// dwords at lane offsets >= `room` come back as 0 (raw buffer range check).
template <bool V16>
struct FastItem;
template <>
struct FastItem<true> {
    typedef f64x2 type;
    static __device__ __forceinline__ type load(uint64_t base, unsigned room, unsigned lane_off) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), (short)0,
                                                            (int)room, 0x00020000);
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, 0, 0);
        return __builtin_bit_cast(f64x2, raw);
    }
};
template <>
struct FastItem<false> {
    typedef double type;
    static __device__ __forceinline__ type load(uint64_t base, unsigned room, unsigned lane_off) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), (short)0,
                                                            (int)room, 0x00020000);
        const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, 0, 0);
        return __builtin_bit_cast(double, raw);
    }
};

// bytes from p to end, saturated to 32 bits (scalar ALU: both operands are wave-uniform)
__device__ __forceinline__ unsigned bytes_left(uint64_t end, uint64_t p) {
    const uint64_t d = end > p ? end - p : 0;
    return d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)d;
}

// Estimated time of qs_gemm_fast.hip's best form for a product (exact form when every extent is a whole number of tiles,
// otherwise the best edge-form shape; `even`: 16-byte aligned bases, even strides and extents), in the units of its shape
// weights: rounds of the tile list over two workgroups per CU x tile area / relative rate.  Host side only.
double gemm_fast_estimate(int dtype, int64_t m, int64_t n, int64_t k, int64_t batch, bool even);

}  // namespace qs
