// Exact three-way bf16 split of fp32 values ("bf16x6" arithmetic, see conv_gemm_bf16.hip): x = hi + mid + lo with
// 8+8+8 significand bits; a*b ~ hh + (hm + mh) + (hl + mm + lh) drops only terms below 2^-24 |a*b|.
#pragma once
#include "agcn_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned sb_pack_bf16(float a, float b) {
  f32x2_t v = {a, b};
  bf16x2_t p = __builtin_convertvector(v, bf16x2_t);      // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float sb_lo_as_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float sb_hi_as_f32(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// (a, b) -> packed bf16 pairs of the three pieces
__device__ __forceinline__ void sb_split_pair(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
  ph = sb_pack_bf16(a, b);
  const float ra = a - sb_lo_as_f32(ph), rb = b - sb_hi_as_f32(ph);
  pm = sb_pack_bf16(ra, rb);
  pl = sb_pack_bf16(ra - sb_lo_as_f32(pm), rb - sb_hi_as_f32(pm));
}

// the products of one 32x32x16 step, smallest first (a*/b*: hi, mid, lo fragments).  npl (wave-uniform): 3 = the six
// products of the fp32-equivalent split, 1 = the hi*hi product only (plain bf16 arithmetic, BASELINE configs[3]).
__device__ __forceinline__ f32x16 sb_mfma6(bf16x8 a0, bf16x8 a1, bf16x8 a2, bf16x8 b0, bf16x8 b1, bf16x8 b2, f32x16 c,
                                           int npl = 3) {
  if (npl == 3) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
  }
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
  return c;
}

template <auto KERN>      // keyed by the kernel itself (instantiations of one template share a function TYPE)
static inline int agcn_allow_big_lds() {
  static unsigned char done[AGCN_MAX_DEVICES] = {};
  return agcn_allow_big_lds_rt(reinterpret_cast<const void*>(KERN), done);
}
