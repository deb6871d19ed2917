// Common device/host helpers for the agcn_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AGCN_OK 0
#define AGCN_ERR_ARG (-1)
#define AGCN_ERR_WORKSPACE (-2)
#define AGCN_ERR_UNSUPPORTED (-3)

// exact-f32 matrix core op: D(32x32) += A(32x2) * B(2x32); lane l holds A[l&31][l>>5], B[l>>5][l&31];
// D register j of lane l is D[row = (j&3) + 8*(j>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

// sum over the 32 lanes of a half-wave (lanes l and l^k stay inside the half for k<32)
__device__ __forceinline__ float half_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// Bijective XCD-aware remap of a 1-D block id (guide T1): blocks b and b+8 share an XCD, so give
// every XCD a contiguous range of logical ids (neighbouring tiles share halos / rows in its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, k = bid >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + k;
}

extern "C" size_t agcn_colsum_scratch_bytes(int W);
extern "C" int agcn_colsum(const float* X, int nslots, int W, void* scratch, float* out, void* stream);

static inline int agcn_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AGCN_OK : (int)e;
}
