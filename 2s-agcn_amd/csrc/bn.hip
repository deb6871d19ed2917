// Train/eval BatchNorm2d + residual + ReLU passes of unit_gcn / unit_tcn / TCN_GCN_unit
// (reference agcn.py:43,49,74,79,107-109,128-129).  HBM-bound elementwise and per-channel reduction kernels on
// (N, C, P=T*V) fp32 tensors; channel statistics come from the partial (sum, sumsq) slabs the contraction
// kernels emit in their epilogue and are combined here in double precision, in a fixed order.
#include "agcn_common.h"

namespace {

constexpr int RG = 64;   // slot groups of the two-stage column reduction

// stage 1 of a column sum X[nslots][W] -> partial[RG][W] (double): grid (ceil(W/64), RG), 64 columns x 4 slot lanes
__global__ void __launch_bounds__(256)
colsum_stage1_kernel(const float* __restrict__ X, int nslots, int W, int per_group, double* __restrict__ partial) {
  __shared__ double red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int g = blockIdx.y;
  const int s_end = min(nslots, (g + 1) * per_group);
  double acc = 0.0;
  if (col < W)
    for (int s = g * per_group + sub; s < s_end; s += 4) acc += (double)X[(long)s * W + col];
  red[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && col < W)
    partial[(long)g * W + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void colsum_stage2_kernel(const double* __restrict__ partial, int W, float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= W) return;
  double s = 0.0;
  for (int g = 0; g < RG; ++g) s += partial[(long)g * W + col];
  out[col] = (float)s;
}

// ---- forward statistics -> mean / invstd / folded affine, running-stat update (momentum, unbiased var) ----
// partial: [RG][2][C] doubles from colsum_stage1 over the [nslots][2][C] slab of the contraction epilogue
__global__ void bn_stats_finalize_kernel(const double* __restrict__ partial, int C, double count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                         float eps, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                         float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, ss = 0.0;
  for (int g = 0; g < RG; ++g) {
    s += partial[((long)g * 2 + 0) * C + c];
    ss += partial[((long)g * 2 + 1) * C + c];
  }
  const double mean = s / count;
  double var = ss / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale_out[c] = sc;
  shift_out[c] = beta[c] - (float)mean * sc;
  if (rmean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_coeff_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int C,
                                     float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rvar[c] + eps);
  scale_out[c] = sc;
  shift_out[c] = beta[c] - rmean[c] * sc;
}

// out = act( scale1[c]*y1 + shift1[c] + res ),  res = 0 | r | scale2[c]*r + shift2[c]
template <int RES, bool RELU>
__global__ void __launch_bounds__(256)
bn_act_fwd_kernel(const float4* __restrict__ y1, const float* __restrict__ scale1, const float* __restrict__ shift1,
                  const float4* __restrict__ r, const float* __restrict__ scale2, const float* __restrict__ shift2,
                  float4* __restrict__ out, unsigned* __restrict__ bits, unsigned total4, unsigned P, unsigned C,
                  unsigned* __restrict__ amax) {
  unsigned tmax = 0;                     // max |out| of this thread (bit pattern: unsigned order = float order)
  // block-uniform trip count: the sign-mask words are assembled with shuffles over groups of 8 lanes (32 elements)
  for (unsigned i0 = blockIdx.x * blockDim.x; i0 < total4; i0 += gridDim.x * blockDim.x) {
    const unsigned i = i0 + threadIdx.x;
    const bool valid = i < total4;
    const unsigned ic = valid ? i : 0u;
    const unsigned e0 = ic * 4u;
    const unsigned row = e0 / P;
    const unsigned rem = e0 - row * P;
    const unsigned c0 = row % C;
    const unsigned c1 = (c0 + 1 == C) ? 0u : c0 + 1;
    const float4 a = y1[ic];
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RES != 0) b = r[ic];
    float av[4] = {a.x, a.y, a.z, a.w};
    float bv[4] = {b.x, b.y, b.z, b.w};
    float ov[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned c = (rem + k < P) ? c0 : c1;
      float v = scale1[c] * av[k] + shift1[c];
      if (RES == 1) v += bv[k];
      if (RES == 2) v += scale2[c] * bv[k] + shift2[c];
      ov[k] = RELU ? fmaxf(v, 0.f) : v;
    }
    if (valid) out[i] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    if (amax && valid) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tmax = max(tmax, __float_as_uint(ov[k]) & 0x7fffffffu);
    }
    if (bits) {
      // bit e of word w <-> element 32*w + e is positive: what every backward pass needs of `out`, 32x smaller
      unsigned nib = valid ? ((ov[0] > 0.f) | ((ov[1] > 0.f) << 1) | ((ov[2] > 0.f) << 2) | ((ov[3] > 0.f) << 3)) : 0u;
      unsigned w = nib << (4u * (threadIdx.x & 7u));
      w |= __shfl_xor(w, 1);
      w |= __shfl_xor(w, 2);
      w |= __shfl_xor(w, 4);
      if ((threadIdx.x & 7u) == 0u && valid) bits[e0 >> 5] = w;
    }
  }
  if (amax) {                            // by-product for the f16x3 kernels that read `out`: the tensor's max |x|
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) tmax = max(tmax, (unsigned)__shfl_xor((int)tmax, k));
    __shared__ unsigned wmax[4];           // one atomic per workgroup (<= 4096 per launch)
    if ((threadIdx.x & 63u) == 0u) wmax[threadIdx.x >> 6] = tmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
      if (m) atomicMax(amax, m);
    }
  }
}

// per (n,c) row: s0 = sum dz, s1 = sum dz*y1, s2 = sum dz*y2, dz = dout * (mask > 0)
template <bool HAS2>
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ mask, int mask_bits,
                     const float* __restrict__ y1, const float* __restrict__ y2, float* __restrict__ part, int P) {
  __shared__ float red[3][4];
  const long row = blockIdx.x;
  const float* d = dout + row * P;
  const float* mk = (mask && !mask_bits) ? mask + row * P : nullptr;
  const unsigned* mb = (mask && mask_bits) ? reinterpret_cast<const unsigned*>(mask) : nullptr;
  const long e_row = row * P;
  const float* a = y1 + row * P;
  const float* b = HAS2 ? y2 + row * P : nullptr;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  // 16-byte loads for every row length: rows of P % 4 != 0 floats (T = 150, 75) start 4- or 8-byte aligned only, which
  // the loads tolerate; the last P % 4 elements go through the scalar tail below
  typedef float4 float4_u __attribute__((aligned(4)));
  const int P4 = P >> 2;
  for (int q4 = threadIdx.x; q4 < P4; q4 += 256) {
    const float4 d4 = reinterpret_cast<const float4_u*>(d)[q4];
    const float4 a4 = reinterpret_cast<const float4_u*>(a)[q4];
    float dv[4] = {d4.x, d4.y, d4.z, d4.w};
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
    if (mk) {
      const float4 m4 = reinterpret_cast<const float4_u*>(mk)[q4];
      const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) dv[k] = (mv[k] > 0.f) ? dv[k] : 0.f;
    }
    if (mb) {
      const long e = e_row + 4L * q4;
      const int sh = (int)(e & 31);
      unsigned nib = mb[e >> 5] >> sh;
      if (sh > 28) nib |= mb[(e >> 5) + 1] << (32 - sh);   // the 4 bits straddle two words (rows not a multiple of 4)
#pragma unroll
      for (int k = 0; k < 4; ++k) dv[k] = ((nib >> k) & 1u) ? dv[k] : 0.f;
    }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS2) {
      const float4 b4 = reinterpret_cast<const float4_u*>(b)[q4];
      bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s0 += dv[k];
      s1 += dv[k] * av[k];
      if (HAS2) s2 += dv[k] * bv[k];
    }
  }
  for (int q = 4 * P4 + threadIdx.x; q < P; q += 256) {
    float dz = d[q];
    if (mk) dz = (mk[q] > 0.f) ? dz : 0.f;
    if (mb) {
      const long e = e_row + q;
      dz = ((mb[e >> 5] >> (e & 31)) & 1u) ? dz : 0.f;
    }
    s0 += dz;
    s1 += dz * a[q];
    if (HAS2) s2 += dz * b[q];
  }
  s0 = half_sum(s0); s1 = half_sum(s1); s2 = half_sum(s2);
  s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    part[row * 3 + k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}

// coef[0..2][C] = (A1,B1,C1) with dy1 = A1*dz + B1*y1 + C1 ; coef[3..5][C] same for branch 2
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int N, int C, double count, float pscale,
                                       const float* __restrict__ gamma1, const float* __restrict__ mean1,
                                       const float* __restrict__ invstd1, const float* __restrict__ gamma2,
                                       const float* __restrict__ mean2, const float* __restrict__ invstd2,
                                       float* __restrict__ coef, float* __restrict__ dgamma1,
                                       float* __restrict__ dbeta1, float* __restrict__ dgamma2,
                                       float* __restrict__ dbeta2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 8
  for (int n = 0; n < N; ++n) {
    const float* p = part + ((long)n * C + c) * 3;
    s0 += (double)p[0]; s1 += (double)p[1]; s2 += (double)p[2];
  }
  {
    const double is = invstd1[c], mu = mean1[c], g = gamma1[c];
    const double sxh = is * (s1 - mu * s0);     // sum dz * xhat
    const double k = g * is;
    coef[0 * C + c] = (float)k;
    coef[1 * C + c] = (float)(-k * is * sxh / count);
    coef[2 * C + c] = (float)(-k * s0 / count + k * is * mu * sxh / count);
    dgamma1[c] = (float)sxh * pscale;
    dbeta1[c] = (float)s0 * pscale;
  }
  if (gamma2) {
    const double is = invstd2[c], mu = mean2[c], g = gamma2[c];
    const double sxh = is * (s2 - mu * s0);
    const double k = g * is;
    coef[3 * C + c] = (float)k;
    coef[4 * C + c] = (float)(-k * is * sxh / count);
    coef[5 * C + c] = (float)(-k * s0 / count + k * is * mu * sxh / count);
    dgamma2[c] = (float)sxh * pscale;
    dbeta2[c] = (float)s0 * pscale;
  }
}

template <bool HAS2>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float4* __restrict__ dout, const float4* __restrict__ mask, int mask_bits,
                    const float4* __restrict__ y1,
                    const float4* __restrict__ y2, const float* __restrict__ coef, float4* __restrict__ dy1,
                    float4* __restrict__ dy2, unsigned total4, unsigned P, unsigned C, unsigned* __restrict__ amax1) {
  unsigned tmax = 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    const unsigned e0 = i * 4u;
    const unsigned row = e0 / P;
    const unsigned rem = e0 - row * P;
    const unsigned c0 = row % C;
    const unsigned c1 = (c0 + 1 == C) ? 0u : c0 + 1;
    const float4 d4 = dout[i];
    const float4 a4 = y1[i];
    float dv[4] = {d4.x, d4.y, d4.z, d4.w};
    float av[4] = {a4.x, a4.y, a4.z, a4.w};
    if (mask && !mask_bits) {
      const float4 m4 = mask[i];
      const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) dv[k] = (mv[k] > 0.f) ? dv[k] : 0.f;
    } else if (mask) {
      const unsigned nib = reinterpret_cast<const unsigned*>(mask)[e0 >> 5] >> (e0 & 31u);
#pragma unroll
      for (int k = 0; k < 4; ++k) dv[k] = ((nib >> k) & 1u) ? dv[k] : 0.f;
    }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS2) {
      const float4 b4 = y2[i];
      bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
    }
    float o1[4], o2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned c = (rem + k < P) ? c0 : c1;
      o1[k] = coef[0 * C + c] * dv[k] + coef[1 * C + c] * av[k] + coef[2 * C + c];
      if (HAS2) o2[k] = coef[3 * C + c] * dv[k] + coef[4 * C + c] * bv[k] + coef[5 * C + c];
    }
    dy1[i] = make_float4(o1[0], o1[1], o1[2], o1[3]);
    if (HAS2) dy2[i] = make_float4(o2[0], o2[1], o2[2], o2[3]);
    if (amax1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tmax = max(tmax, __float_as_uint(o1[k]) & 0x7fffffffu);
    }
  }
  if (amax1) {                           // max |dy1| for the f16x3 kernels that read it
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) tmax = max(tmax, (unsigned)__shfl_xor((int)tmax, k));
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63u) == 0u) wmax[threadIdx.x >> 6] = tmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
      if (m) atomicMax(amax1, m);
    }
  }
}

inline unsigned ew_grid(unsigned total4) {
  unsigned g = (total4 + 255u) / 256u;
  return g > 4096u ? 4096u : (g ? g : 1u);
}

}  // namespace

extern "C" {

// doubles of scratch the two-stage column reductions need for a width-W slab
size_t agcn_colsum_scratch_bytes(int W) { return sizeof(double) * (size_t)RG * (size_t)W; }

// out[w] = sum_s X[s][w], fixed summation order (bitwise reproducible), double accumulation
int agcn_colsum(const float* X, int nslots, int W, void* scratch, float* out, void* stream) {
  if (!X || !scratch || !out || nslots <= 0 || W <= 0) return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int per_group = (nslots + RG - 1) / RG;
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((W + 63) / 64, RG), dim3(256), 0, s, X, nslots, W, per_group,
                     (double*)scratch);
  int rc = agcn_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((W + 255) / 256), dim3(256), 0, s, (const double*)scratch, W, out);
  return agcn_check_launch();
}

// part: [nslots][2][C] from the contraction epilogue ; count = N*T*V ; rmean/rvar may be null (no update);
// scratch: agcn_colsum_scratch_bytes(2*C) bytes
int agcn_bn_stats_finalize(const float* part, int nslots, int C, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, void* scratch,
                           float* mean, float* invstd, float* scale, float* shift, void* stream) {
  if (!part || !gamma || !beta || !scratch || !mean || !invstd || !scale || !shift || C <= 0 || nslots <= 0)
    return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int W = 2 * C;
  const int per_group = (nslots + RG - 1) / RG;
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((W + 63) / 64, RG), dim3(256), 0, s, part, nslots, W, per_group,
                     (double*)scratch);
  int rc = agcn_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, s, (const double*)scratch, C, count,
                     gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
  return agcn_check_launch();
}

int agcn_bn_eval_coeff(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int C, float* scale, float* shift, void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return AGCN_ERR_ARG;
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  return agcn_check_launch();
}

// res_mode: 0 none, 1 identity residual r, 2 BN'd residual branch scale2*r+shift2 ; total = N*C*P must be %4
int agcn_bn_act_fwd_ex(const float* y1, const float* scale1, const float* shift1, const float* r, const float* scale2,
                       const float* shift2, float* out, unsigned* sign_bits, float* absmax_out, int N, int C, int P,
                       int res_mode, int relu, void* stream);
int agcn_bn_act_fwd(const float* y1, const float* scale1, const float* shift1, const float* r, const float* scale2,
                    const float* shift2, float* out, unsigned* sign_bits, int N, int C, int P, int res_mode, int relu,
                    void* stream) {
  return agcn_bn_act_fwd_ex(y1, scale1, shift1, r, scale2, shift2, out, sign_bits, nullptr, N, C, P, res_mode, relu, stream);
}

// same; absmax_out (optional, 4 bytes): max |out| as a float, for the split-fp16 kernels that read `out` next
int agcn_bn_act_fwd_ex(const float* y1, const float* scale1, const float* shift1, const float* r, const float* scale2,
                       const float* shift2, float* out, unsigned* sign_bits, float* absmax_out, int N, int C, int P,
                       int res_mode, int relu, void* stream) {
  if (!y1 || !scale1 || !shift1 || !out || N <= 0 || C <= 0 || P <= 0) return AGCN_ERR_ARG;
  if (absmax_out && hipMemsetAsync(absmax_out, 0, 4, (hipStream_t)stream) != hipSuccess) return AGCN_ERR_ARG;
  const long total = (long)N * C * P;
  if (total % 4 != 0 || total > 0xffffffffL) return AGCN_ERR_UNSUPPORTED;   // (element indices are 32-bit in the kernels)
  if (res_mode != 0 && !r) return AGCN_ERR_ARG;
  if (res_mode == 2 && (!scale2 || !shift2)) return AGCN_ERR_ARG;
  const unsigned t4 = (unsigned)(total / 4);
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(ew_grid(t4)), b(256);
#define LAUNCH_ACT(R, A)                                                                                       \
  hipLaunchKernelGGL((bn_act_fwd_kernel<R, A>), g, b, 0, s, (const float4*)y1, scale1, shift1, (const float4*)r, \
                     scale2, shift2, (float4*)out, sign_bits, t4, (unsigned)P, (unsigned)C, (unsigned*)absmax_out)
  if (relu) {
    if (res_mode == 0) LAUNCH_ACT(0, true); else if (res_mode == 1) LAUNCH_ACT(1, true); else LAUNCH_ACT(2, true);
  } else {
    if (res_mode == 0) LAUNCH_ACT(0, false); else if (res_mode == 1) LAUNCH_ACT(1, false); else LAUNCH_ACT(2, false);
  }
#undef LAUNCH_ACT
  return agcn_check_launch();
}

// Backward of out = relu(bn1(y1) + [bn2(y2)] + ...): dz = dout*(mask>0) (mask may be null);
// dy1 = A1*dz + B1*y1 + C1 (train-mode BN backward), same for branch 2.  Two stages so that a synchronised BatchNorm
// (reference DDP + SyncBatchNorm, utils/processor.py:295) can all-reduce the per-channel sums in between:
//   agcn_bn_bwd_reduce: part[(n*C + c)*3 + k] = per-row (sum dz, sum dz*y1, sum dz*y2)
//   agcn_bn_bwd_apply : sums `nrows` rows of part per channel (fixed order), then the coefficients over `count`
//                       elements and the element-wise pass; dgamma/dbeta are multiplied by param_grad_scale
//                       (1/world when the sums are global, so that the gradient all-reduce average restores them).
int agcn_bn_bwd_reduce(const float* dout, const void* mask, int mask_bits, const float* y1, const float* y2, float* part,
                       int N, int C, int P, void* stream) {
  if (!dout || !y1 || !part || N <= 0 || C <= 0 || P <= 0) return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (y2) hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(N * C), dim3(256), 0, s, dout, (const float*)mask,
                             mask_bits, y1, y2, part, P);
  else hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(N * C), dim3(256), 0, s, dout, (const float*)mask,
                          mask_bits, y1, y2, part, P);
  return agcn_check_launch();
}

int agcn_bn_bwd_apply_ex(const float* part, int nrows, double count, float param_grad_scale, const float* dout,
                         const void* mask, int mask_bits, const float* y1, const float* gamma1, const float* mean1,
                         const float* invstd1, const float* y2, const float* gamma2, const float* mean2,
                         const float* invstd2, float* coef, float* dy1, float* dgamma1, float* dbeta1, float* dy2,
                         float* dgamma2, float* dbeta2, float* absmax1_out, int N, int C, int P, void* stream);
int agcn_bn_bwd_apply(const float* part, int nrows, double count, float param_grad_scale, const float* dout,
                      const void* mask, int mask_bits, const float* y1, const float* gamma1, const float* mean1, const float* invstd1,
                      const float* y2, const float* gamma2, const float* mean2, const float* invstd2, float* coef,
                      float* dy1, float* dgamma1, float* dbeta1, float* dy2, float* dgamma2, float* dbeta2, int N, int C,
                      int P, void* stream) {
  return agcn_bn_bwd_apply_ex(part, nrows, count, param_grad_scale, dout, mask, mask_bits, y1, gamma1, mean1, invstd1, y2,
                              gamma2, mean2, invstd2, coef, dy1, dgamma1, dbeta1, dy2, dgamma2, dbeta2, nullptr, N, C, P,
                              stream);
}

// same; absmax1_out (optional, 4 bytes): max |dy1| as a float, for the split-fp16 kernels that read dy1 next
int agcn_bn_bwd_apply_ex(const float* part, int nrows, double count, float param_grad_scale, const float* dout,
                         const void* mask, int mask_bits, const float* y1, const float* gamma1, const float* mean1,
                         const float* invstd1, const float* y2, const float* gamma2, const float* mean2,
                         const float* invstd2, float* coef, float* dy1, float* dgamma1, float* dbeta1, float* dy2,
                         float* dgamma2, float* dbeta2, float* absmax1_out, int N, int C, int P, void* stream) {
  if (absmax1_out && hipMemsetAsync(absmax1_out, 0, 4, (hipStream_t)stream) != hipSuccess) return AGCN_ERR_ARG;
  if (!part || !dout || !y1 || !gamma1 || !mean1 || !invstd1 || !coef || !dy1 || !dgamma1 || !dbeta1 || nrows <= 0)
    return AGCN_ERR_ARG;
  if (y2 && (!gamma2 || !mean2 || !invstd2 || !dy2 || !dgamma2 || !dbeta2)) return AGCN_ERR_ARG;
  const long total = (long)N * C * P;
  if (total % 4 != 0 || total > 0xffffffffL) return AGCN_ERR_UNSUPPORTED;   // (element indices are 32-bit in the kernels)
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, s, part, nrows, C, count,
                     param_grad_scale, gamma1, mean1, invstd1, y2 ? gamma2 : nullptr, mean2, invstd2, coef, dgamma1,
                     dbeta1, dgamma2, dbeta2);
  int rc = agcn_check_launch();
  if (rc) return rc;
  const unsigned t4 = (unsigned)(total / 4);
  const dim3 g(ew_grid(t4)), b(256);
  if (y2)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, g, b, 0, s, (const float4*)dout, (const float4*)mask, mask_bits,
                       (const float4*)y1, (const float4*)y2, (const float*)coef, (float4*)dy1, (float4*)dy2, t4,
                       (unsigned)P, (unsigned)C, (unsigned*)absmax1_out);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, g, b, 0, s, (const float4*)dout, (const float4*)mask, mask_bits,
                       (const float4*)y1, (const float4*)y2, (const float*)coef, (float4*)dy1, (float4*)dy2, t4,
                       (unsigned)P, (unsigned)C, (unsigned*)absmax1_out);
  return agcn_check_launch();
}

// both stages back to back (per-replica statistics).  part: (N*C*3) scratch, coef: (6*C) scratch.
int agcn_bn_bwd(const float* dout, const void* mask, int mask_bits, const float* y1, const float* gamma1,
                const float* mean1, const float* invstd1, const float* y2, const float* gamma2, const float* mean2,
                const float* invstd2, float* part, float* coef, float* dy1, float* dgamma1, float* dbeta1, float* dy2,
                float* dgamma2, float* dbeta2, int N, int C, int P, void* stream) {
  int rc = agcn_bn_bwd_reduce(dout, mask, mask_bits, y1, y2, part, N, C, P, stream);
  if (rc) return rc;
  return agcn_bn_bwd_apply(part, N, (double)N * (double)P, 1.0f, dout, mask, mask_bits, y1, gamma1, mean1, invstd1, y2, gamma2,
                           mean2, invstd2, coef, dy1, dgamma1, dbeta1, dy2, dgamma2, dbeta2, N, C, P, stream);
}

const char* agcn_arch(void) { return "gfx950"; }
int agcn_version(void) { return 100; }

}  // extern "C"
