// The small operators AROUND the TCN_GCN_unit stack, as deterministic hand-written kernels (no vendor library, no float
// atomics, every sum in a fixed order), so that the whole model path is bitwise reproducible from process to process:
//
//   * data_bn (reference agcn.py:143,163-165 / aagcn.py forward_preprocess): BatchNorm1d over channels (m, v, c) of the
//     input (N, C, T, V, M), statistics over (N, T), fused with the two permutes -> (N*M, C, T, V);
//   * global average pool + classifier (agcn.py:179-183): mean over (T, V) per person, mean over persons, Linear;
//   * AAGCN's gate networks (aagcn.py:72-76, 92-96, 111-116): Conv1d(C -> 1, k, 'same') + sigmoid on (N, C, L) and the
//     Linear -> ReLU -> Linear -> sigmoid pair on (N, C).
//
// Why they exist: MIOpen's Conv1d picked for the temporal gate at C = 128 gave different bits from process to process on
// the same box with identical inputs (tools/stage_checksums.py), which made every downstream tensor
// of an AAGCN run irreproducible.  All tensors here are a few KB to a few MB; the kernels aim at "a handful of
// microseconds", not at a roofline.
#include "agcn_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// data_bn
// ---------------------------------------------------------------------------------------------------------------------
// x (N, C, T, V, M); channel of element (c, v, m): ch = (m*V + v)*C + c   (the reference's permute(0,4,3,1,2).view)
// stage 1: one workgroup per (n, c): thread j <-> (v, m) pair (j = v*M + m, contiguous in memory), loop over t.
// part: [N][2][CH]
__global__ void __launch_bounds__(64) data_bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int N,
                                                           int C, int T, int V, int M) {
  const int n = blockIdx.x / C, c = blockIdx.x - n * C;
  const int VM = V * M, CH = C * VM;
  for (int j = threadIdx.x; j < VM; j += 64) {
    const float* p = x + ((long)(n * C + c) * T) * VM + j;
    float s = 0.f, ss = 0.f;
    for (int t = 0; t < T; ++t) {
      const float v = p[(long)t * VM];
      s += v;
      ss += v * v;
    }
    const int v_ = j / M, m = j - v_ * M;
    const int ch = (m * V + v_) * C + c;
    part[((long)n * 2 + 0) * CH + ch] = s;
    part[((long)n * 2 + 1) * CH + ch] = ss;
  }
}

// out (N*M, C, T, V)[(n*M+m), c, t, v] = x[n, c, t, v, m] * scale[ch] + shift[ch]
__global__ void __launch_bounds__(256) data_bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            int N, int C, int T, int V, int M) {
  const long total = (long)N * M * C * T * V;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int v = (int)(e % V);
    long r = e / V;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % C); r /= C;
    const int m = (int)(r % M);
    const int n = (int)(r / M);
    const int ch = (m * V + v) * C + c;
    out[e] = x[((((long)n * C + c) * T + t) * V + v) * M + m] * scale[ch] + shift[ch];
  }
}

// backward stage 1: one workgroup per output row (n*M+m, c): thread <-> v, loop over t
// part: [N*M][2][CH-compatible]: slot s = n*M+m holds only its own m's channels (others zero), so a plain column sum over
// the slots gives (sum dy, sum dy*xhat) per channel
__global__ void __launch_bounds__(64) data_bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ part,
                                                                int N, int C, int T, int V, int M) {
  const int row = blockIdx.x / C, c = blockIdx.x - row * C;     // row = n*M + m
  const int n = row / M, m = row - n * M;
  const int CH = C * V * M;
  const int v = threadIdx.x;
  if (v >= V) return;
  const int ch = (m * V + v) * C + c;
  const float mu = mean[ch], is = invstd[ch];
  const float* dp = dy + ((long)row * C + c) * T * V + v;
  const float* xp = x + (((long)n * C + c) * T) * V * M + (long)v * M + m;
  float s = 0.f, sx = 0.f;
  for (int t = 0; t < T; ++t) {
    const float g = dp[(long)t * V];
    const float xh = (xp[(long)t * V * M] - mu) * is;
    s += g;
    sx += g * xh;
  }
  // slab [N][2][CH]: every (n, ch) written exactly once (ch identifies m)
  part[((long)n * 2 + 0) * CH + ch] = s;
  part[((long)n * 2 + 1) * CH + ch] = sx;
}

// dx[n,c,t,v,m] = gamma*invstd * (dy - sum_dy/count - xhat * sum_dyxhat/count)
__global__ void __launch_bounds__(256) data_bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ sums, float inv_count,
                                                                float* __restrict__ dx, int N, int C, int T, int V, int M) {
  const long total = (long)N * M * C * T * V;
  const int CH = C * V * M;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int v = (int)(e % V);
    long r = e / V;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % C); r /= C;
    const int m = (int)(r % M);
    const int n = (int)(r / M);
    const int ch = (m * V + v) * C + c;
    const long xi = ((((long)n * C + c) * T + t) * V + v) * M + m;
    const float xh = (x[xi] - mean[ch]) * invstd[ch];
    dx[xi] = gamma[ch] * invstd[ch] * (dy[e] - sums[ch] * inv_count - xh * sums[CH + ch] * inv_count);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// global average pool over (T, V) then over persons
// ---------------------------------------------------------------------------------------------------------------------
// rowmean[(n*M+m), c] = mean_p x[row, c, p]: one workgroup per row, fixed-order tree
__global__ void __launch_bounds__(256) pool_rows_kernel(const float* __restrict__ x, float* __restrict__ rowmean, int P) {
  __shared__ float red[256];
  const float* p = x + (long)blockIdx.x * P;
  float s = 0.f;
  for (int i = threadIdx.x; i < P; i += 256) s += p[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowmean[blockIdx.x] = red[0] / (float)P;
}
// pooled[n, c] = mean_m rowmean[(n*M+m), c]
__global__ void pool_persons_kernel(const float* __restrict__ rowmean, float* __restrict__ pooled, int N, int M, int C) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * C) return;
  const int n = e / C, c = e - n * C;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += rowmean[((long)n * M + m) * C + c];
  pooled[e] = s / (float)M;
}
// dx[(n*M+m), c, p] = dpooled[n, c] / (M*P)
__global__ void __launch_bounds__(256) pool_bwd_kernel(const float* __restrict__ dpooled, float* __restrict__ dx, int N,
                                                       int M, int C, int P, float scale) {
  const int row = blockIdx.x;                 // (n*M+m)*C + c
  const int c = row % C, n = (row / C) / M;
  const float g = dpooled[(long)n * C + c] * scale;
  float* p = dx + (long)row * P;
  for (int i = threadIdx.x; i < P; i += 256) p[i] = g;
}

// ---------------------------------------------------------------------------------------------------------------------
// small Linear: out[n, o] = act(b[o] + sum_k in[n, k] * w[o, k]);  act 0 identity, 1 ReLU, 2 1 + sigmoid
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 1.f + 1.f / (1.f + expf(-v));
  return v;
}
// derivative expressed through the OUTPUT: relu: out > 0 ; 1+sigmoid: s = out-1, s(1-s)
__device__ __forceinline__ float act_grad(float out, int act) {
  if (act == 1) return out > 0.f ? 1.f : 0.f;
  if (act == 2) return (out - 1.f) * (2.f - out);
  return 1.f;
}

// one workgroup per sample n: the input row is staged in LDS, thread <-> output o
__global__ void __launch_bounds__(256) linear_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ out, int K, int O,
                                                         int act) {
  extern __shared__ float row[];
  const int n = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += 256) row[k] = in[(long)n * K + k];
  __syncthreads();
  for (int o = threadIdx.x; o < O; o += 256) {
    const float* wr = w + (long)o * K;
    float s = b ? b[o] : 0.f;
    for (int k = 0; k < K; ++k) s += row[k] * wr[k];
    out[(long)n * O + o] = act_apply(s, act);
  }
}
// dpre[n, o] = dout * act'(out) ; din[n, k] = sum_o dpre[n, o] * w[o, k]   (one workgroup per n)
__global__ void __launch_bounds__(256) linear_bwd_in_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                            const float* __restrict__ w, float* __restrict__ dpre,
                                                            float* __restrict__ din, int K, int O, int act) {
  extern __shared__ float dp[];
  const int n = blockIdx.x;
  for (int o = threadIdx.x; o < O; o += 256) {
    const float g = dout[(long)n * O + o] * act_grad(out[(long)n * O + o], act);
    dp[o] = g;
    dpre[(long)n * O + o] = g;
  }
  __syncthreads();
  if (!din) return;
  for (int k = threadIdx.x; k < K; k += 256) {
    float s = 0.f;
    for (int o = 0; o < O; ++o) s += dp[o] * w[(long)o * K + k];
    din[(long)n * K + k] = s;
  }
}
// dw[o, k] = sum_n dpre[n, o] * in[n, k] ; db[o] = sum_n dpre[n, o]   (thread per weight element, n in order)
__global__ void __launch_bounds__(256) linear_bwd_w_kernel(const float* __restrict__ dpre, const float* __restrict__ in,
                                                           float* __restrict__ dw, float* __restrict__ db, int N, int K,
                                                           int O) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < O * K) {
    const int o = e / K, k = e - o * K;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dpre[(long)n * O + o] * in[(long)n * K + k];
    dw[e] = s;
  }
  if (db && e < O) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dpre[(long)n * O + e];
    db[e] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// gate convolution: a[n, l] = 1 + sigmoid(b + sum_c sum_k w[c, k] * in[n, c, l + k - pad]),  pad = (Ks-1)/2
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GC_LT = 32, GC_CP = 8;   // positions per workgroup, channel partitions (combined in a fixed order)

__global__ void __launch_bounds__(256) gate_conv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ out, int C,
                                                            int L, int Ks) {
  __shared__ float red[GC_CP][GC_LT];
  const int n = blockIdx.y, l0 = blockIdx.x * GC_LT;
  const int li = threadIdx.x % GC_LT, cp = threadIdx.x / GC_LT;
  const int l = l0 + li, pad = (Ks - 1) / 2;
  const int cper = (C + GC_CP - 1) / GC_CP;
  float s = 0.f;
  if (l < L) {
    for (int c = cp * cper; c < min(C, (cp + 1) * cper); ++c) {
      const float* ir = in + ((long)n * C + c) * L;
      const float* wr = w + (long)c * Ks;
      for (int k = 0; k < Ks; ++k) {
        const int q = l + k - pad;
        if (q >= 0 && q < L) s += wr[k] * ir[q];
      }
    }
  }
  red[cp][li] = s;
  __syncthreads();
  if (cp == 0 && l < L) {
    float t = b ? b[0] : 0.f;
#pragma unroll
    for (int p = 0; p < GC_CP; ++p) t += red[p][li];
    out[(long)n * L + l] = 1.f + 1.f / (1.f + expf(-t));
  }
}
// dpre[n, l] = dout * (a-1)(2-a)
__global__ void gate_dpre_kernel(const float* __restrict__ dout, const float* __restrict__ a, float* __restrict__ dpre,
                                 long total) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) dpre[e] = dout[e] * (a[e] - 1.f) * (2.f - a[e]);
}
// din[n, c, l] = sum_k w[c, k] * dpre[n, l - k + pad]
__global__ void __launch_bounds__(256) gate_conv_bwd_in_kernel(const float* __restrict__ dpre, const float* __restrict__ w,
                                                               float* __restrict__ din, int N, int C, int L, int Ks) {
  const long total = (long)N * C * L;
  const int pad = (Ks - 1) / 2;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int l = (int)(e % L);
    const long r = e / L;
    const int c = (int)(r % C), n = (int)(r / C);
    const float* dr = dpre + (long)n * L;
    const float* wr = w + (long)c * Ks;
    float s = 0.f;
    for (int k = 0; k < Ks; ++k) {
      const int q = l - k + pad;
      if (q >= 0 && q < L) s += wr[k] * dr[q];
    }
    din[e] = s;
  }
}
// dw[c, k] = sum_{n, l} dpre[n, l] * in[n, c, l + k - pad] : one workgroup per (c, k), fixed-order tree over 256 threads
__global__ void __launch_bounds__(256) gate_conv_bwd_w_kernel(const float* __restrict__ dpre, const float* __restrict__ in,
                                                              float* __restrict__ dw, int N, int C, int L, int Ks) {
  __shared__ float red[256];
  const int c = blockIdx.x / Ks, k = blockIdx.x - c * Ks;
  const int pad = (Ks - 1) / 2;
  float s = 0.f;
  const int NL = N * L;
  for (int e = threadIdx.x; e < NL; e += 256) {
    const int n = e / L, l = e - n * L;
    const int q = l + k - pad;
    if (q >= 0 && q < L) s += dpre[e] * in[((long)n * C + c) * L + q];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) dw[blockIdx.x] = red[0];
}
// out[0] = sum of `total` floats (one workgroup, fixed order)
__global__ void __launch_bounds__(256) sum_all_kernel(const float* __restrict__ x, float* __restrict__ out, long total) {
  __shared__ float red[256];
  float s = 0.f;
  for (long e = threadIdx.x; e < total; e += 256) s += x[e];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

// ---- data_bn -------------------------------------------------------------------------------------------------------
// part: [N][2][C*V*M] partial (sum, sumsq) per channel ch = (m*V+v)*C + c; finalize with agcn_bn_stats_finalize
// (nslots = N, C = C*V*M, count = N*T)
int agcn_data_bn_stats(const float* x, float* part, int N, int C, int T, int V, int M, void* stream) {
  if (!x || !part || N <= 0 || C <= 0 || T <= 0 || V <= 0 || M <= 0) return AGCN_ERR_ARG;
  hipLaunchKernelGGL(data_bn_stats_kernel, dim3(N * C), dim3(64), 0, (hipStream_t)stream, x, part, N, C, T, V, M);
  return agcn_check_launch();
}

int agcn_data_bn_apply(const float* x, const float* scale, const float* shift, float* out, int N, int C, int T, int V,
                       int M, void* stream) {
  if (!x || !scale || !shift || !out || N <= 0 || C <= 0 || T <= 0 || V <= 0 || M <= 0) return AGCN_ERR_ARG;
  const long total = (long)N * M * C * T * V;
  hipLaunchKernelGGL(data_bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out,
                     N, C, T, V, M);
  return agcn_check_launch();
}

// part: [N][2][C*V*M] partial (sum dy, sum dy*xhat); reduce with agcn_colsum(part, N, 2*C*V*M)
int agcn_data_bn_bwd_reduce(const float* dy, const float* x, const float* mean, const float* invstd, float* part, int N,
                            int C, int T, int V, int M, void* stream) {
  if (!dy || !x || !mean || !invstd || !part || N <= 0 || C <= 0 || T <= 0 || V <= 0 || V > 64 || M <= 0)
    return AGCN_ERR_ARG;
  hipLaunchKernelGGL(data_bn_bwd_reduce_kernel, dim3(N * M * C), dim3(64), 0, (hipStream_t)stream, dy, x, mean, invstd,
                     part, N, C, T, V, M);
  return agcn_check_launch();
}

// sums: [2][C*V*M] = (sum dy, sum dy*xhat) over the GLOBAL batch; count = elements per channel behind them
int agcn_data_bn_bwd_apply(const float* dy, const float* x, const float* gamma, const float* mean, const float* invstd,
                           const float* sums, double count, float* dx, int N, int C, int T, int V, int M, void* stream) {
  if (!dy || !x || !gamma || !mean || !invstd || !sums || !dx || count <= 0 || N <= 0 || C <= 0 || T <= 0 || V <= 0 ||
      M <= 0)
    return AGCN_ERR_ARG;
  const long total = (long)N * M * C * T * V;
  hipLaunchKernelGGL(data_bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean,
                     invstd, sums, (float)(1.0 / count), dx, N, C, T, V, M);
  return agcn_check_launch();
}

// ---- global average pool -------------------------------------------------------------------------------------------
// x (N*M, C, P) -> pooled (N, C) = mean over persons of the mean over positions (agcn.py:179-181);
// rowmean: scratch of N*M*C floats
int agcn_pool_fwd(const float* x, float* rowmean, float* pooled, int N, int M, int C, int P, void* stream) {
  if (!x || !rowmean || !pooled || N <= 0 || M <= 0 || C <= 0 || P <= 0) return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pool_rows_kernel, dim3(N * M * C), dim3(256), 0, s, x, rowmean, P);
  int rc = agcn_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(pool_persons_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, rowmean, pooled, N, M, C);
  return agcn_check_launch();
}

int agcn_pool_bwd(const float* dpooled, float* dx, int N, int M, int C, int P, void* stream) {
  if (!dpooled || !dx || N <= 0 || M <= 0 || C <= 0 || P <= 0) return AGCN_ERR_ARG;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(N * M * C), dim3(256), 0, (hipStream_t)stream, dpooled, dx, N, M, C, P,
                     1.f / ((float)M * (float)P));
  return agcn_check_launch();
}

// ---- small Linear (classifier agcn.py:183; channel-attention pair aagcn.py:111-116) -----------------------------------
int agcn_linear_fwd(const float* in, const float* w, const float* b, float* out, int N, int K, int O, int act,
                    void* stream) {
  if (!in || !w || !out || N <= 0 || K <= 0 || O <= 0 || act < 0 || act > 2 || K > 16384) return AGCN_ERR_ARG;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(N), dim3(256), (size_t)K * 4, (hipStream_t)stream, in, w, b, out, K, O, act);
  return agcn_check_launch();
}

// dpre: scratch (N, O); din may be null; db may be null
int agcn_linear_bwd(const float* dout, const float* out, const float* in, const float* w, float* dpre, float* din,
                    float* dw, float* db, int N, int K, int O, int act, void* stream) {
  if (!dout || !out || !in || !w || !dpre || !dw || N <= 0 || K <= 0 || O <= 0 || act < 0 || act > 2 || O > 16384)
    return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_bwd_in_kernel, dim3(N), dim3(256), (size_t)O * 4, s, dout, out, w, dpre, din, K, O, act);
  int rc = agcn_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((O * K + 255) / 256), dim3(256), 0, s, dpre, in, dw, db, N, K, O);
  return agcn_check_launch();
}

// ---- gate convolution (aagcn.py:72-76, 92-96) ---------------------------------------------------------------------------
// in (N, C, L), w (C, Ks) [= Conv1d weight (1, C, Ks)], b (1) -> a (N, L) = 1 + sigmoid(conv)
int agcn_gate_conv_fwd(const float* in, const float* w, const float* b, float* a, int N, int C, int L, int Ks,
                       void* stream) {
  if (!in || !w || !a || N <= 0 || C <= 0 || L <= 0 || Ks <= 0 || (Ks & 1) == 0) return AGCN_ERR_ARG;
  hipLaunchKernelGGL(gate_conv_fwd_kernel, dim3((L + GC_LT - 1) / GC_LT, N), dim3(256), 0, (hipStream_t)stream, in, w, b,
                     a, C, L, Ks);
  return agcn_check_launch();
}

// da (N, L) gradient w.r.t. a ; dpre: scratch (N, L) ; din (N, C, L), dw (C, Ks), db (1)
int agcn_gate_conv_bwd(const float* da, const float* a, const float* in, const float* w, float* dpre, float* din,
                       float* dw, float* db, int N, int C, int L, int Ks, void* stream) {
  if (!da || !a || !in || !w || !dpre || !din || !dw || !db || N <= 0 || C <= 0 || L <= 0 || Ks <= 0 || (Ks & 1) == 0)
    return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long nl = (long)N * L;
  hipLaunchKernelGGL(gate_dpre_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, da, a, dpre, nl);
  int rc = agcn_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(gate_conv_bwd_in_kernel, dim3(grid_for(nl * C)), dim3(256), 0, s, dpre, w, din, N, C, L, Ks);
  if ((rc = agcn_check_launch())) return rc;
  hipLaunchKernelGGL(gate_conv_bwd_w_kernel, dim3(C * Ks), dim3(256), 0, s, dpre, in, dw, N, C, L, Ks);
  if ((rc = agcn_check_launch())) return rc;
  hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(256), 0, s, dpre, db, nl);
  return agcn_check_launch();
}

}  // extern "C"
