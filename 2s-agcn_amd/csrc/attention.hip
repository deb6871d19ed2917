// AAGCN's spatial / temporal / channel attention gates (reference aagcn.py:59-116, applied :268-270):
//     y1 = y  * (1 + se_s[n,v]),  se_s = sigmoid(conv1d_V(mean_t y))
//     y2 = y1 * (1 + se_t[n,t]),  se_t = sigmoid(conv1d_9(mean_v y1))
//     y3 = y2 * (1 + se_c[n,c]),  se_c = sigmoid(fc2(relu(fc1(mean_{t,v} y2))))
// are bandwidth-bound passes over (N, C, T, V).  The full-tensor work is three kernels here; the few-KB gate networks
// between them (Conv1d C->1, two Linears, sigmoids on (N,C,V) / (N,C,T) / (N,C) tensors) stay ordinary tensor code in
// the host module.
//
//   agcn_stc_row_reduce : one workgroup per (n, c) row (T*V floats, staged ONCE in LDS with coalesced loads):
//        e[t,v]   = y[t,v] * (g ? g[t,v] : 1)
//        out_t[t] = scale_t * sum_v wv[n][v] * e[t,v]          (lane <-> t, LDS walk over v: stride V, conflict-free)
//        out_v[v] = scale_v * sum_t wt[.][t] * e[t,v]          (thread <-> (v, t-slice), fixed-order LDS combine)
//     forward : mean_t y (out_v), mean_v y*(1+se_s) (out_t)
//     backward: P1[n,c,t] = sum_v a_s[v] dout*y, P2[n,c,v] = sum_t a_t[t] dout*y in ONE pass over (dout, y); and
//               R[n,c,v] = sum_t dmv1[n,c,t] y[t,v]
//   agcn_stc_apply      : out = y * a_s[n,v] * a_t[n,t] * a_c[n,c]                        (one read, one write)
//   agcn_stc_bwd_apply  : dy  = dout * a_s a_t a_c + dmv[n,c,t] * a_s[n,v] + dms[n,c,v]  (means' gradients folded in)
// mean_{t,v} y2 needs no pass of its own: it is mean_t( a_t[t] * mean_v(y1)[t] ).
#include "agcn_common.h"

namespace {

struct RowArgs {
  const float* y;
  const float* g;      // optional elementwise factor (dout)
  const float* wv;     // (N, V) or null (ones)
  const float* wt;     // (N, T) or (N*C, T) or null (ones)
  float* out_t;        // (N*C, T) or null
  float* out_v;        // (N*C, V) or null
  float scale_t, scale_v;
  int N, C, T, V, wt_per_row;
};

__global__ void __launch_bounds__(256) stc_row_reduce_kernel(const RowArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = a.T, V = a.V, P = T * V;
  float* row = smem;                    // [P]
  float* wvs = smem + P;                // [32]
  float* wts = wvs + 32;                // [T]
  float* red = wts + T;                 // [8][32] partial column sums
  const int r = blockIdx.x, n = r / a.C, tid = threadIdx.x;
  const float* yr = a.y + (long)r * P;
  const float* gr = a.g ? a.g + (long)r * P : nullptr;
  if ((P & 3) == 0) {                   // rows are 16-byte aligned: float4 loads
    const f32x4* y4 = reinterpret_cast<const f32x4*>(yr);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gr);
    for (int i = tid; i < (P >> 2); i += 256) {
      f32x4 v = y4[i];
      if (gr) { const f32x4 w = g4[i]; v[0] *= w[0]; v[1] *= w[1]; v[2] *= w[2]; v[3] *= w[3]; }
      *reinterpret_cast<f32x4*>(row + 4 * i) = v;
    }
  } else {
    for (int i = tid; i < P; i += 256) row[i] = gr ? yr[i] * gr[i] : yr[i];
  }
  if (tid < 32) wvs[tid] = (tid < V) ? (a.wv ? a.wv[n * V + tid] : 1.f) : 0.f;
  if (a.out_v) {
    const float* w = a.wt ? a.wt + (long)(a.wt_per_row ? r : n) * T : nullptr;
    for (int t = tid; t < T; t += 256) wts[t] = w ? w[t] : 1.f;
  }
  __syncthreads();
  if (a.out_t) {
    for (int t = tid; t < T; t += 256) {
      const float* e = row + t * V;
      float s = 0.f;
      for (int v = 0; v < V; ++v) s += wvs[v] * e[v];
      a.out_t[(long)r * T + t] = s * a.scale_t;
    }
  }
  if (a.out_v) {
    // thread <-> (column v = tid & 31, t-slice sl = tid >> 5): 8 slices, combined in slice order (reproducible)
    const int v = tid & 31, sl = tid >> 5;
    float s = 0.f;
    if (v < V)
      for (int t = sl; t < T; t += 8) s += wts[t] * row[t * V + v];
    red[sl * 32 + v] = s;
    __syncthreads();
    if (tid < V) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += red[k * 32 + tid];
      a.out_v[(long)r * V + tid] = acc * a.scale_v;
    }
  }
}

struct ApplyArgs {
  const float* y;      // fwd: y ; bwd: dout
  const float* as;     // (N, V)  1 + se_s
  const float* at;     // (N, T)  1 + se_t
  const float* ac;     // (N, C)  1 + se_c
  const float* dmv;    // bwd: (N*C, T) gradient of mean_v(y1), already divided by V
  const float* dms;    // bwd: (N*C, V) gradient of mean_t(y),  already divided by T
  float* out;
  long total;
  int C, T, V, bwd;
  unsigned* amax;      // optional: receives max |out| (bit pattern of a non-negative float), zeroed by the launcher
};

__global__ void __launch_bounds__(256) stc_apply_kernel(const ApplyArgs a) {
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  unsigned mx = 0;
  if (i4 < a.total) {
  const int P = a.T * a.V;
  const f32x4 x = *reinterpret_cast<const f32x4*>(a.y + i4);
  f32x4 o;
  long r = i4 / P;
  int p = (int)(i4 - r * P);
  int t = p / a.V, v = p - t * a.V;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n = (int)(r / a.C), c = (int)(r - (long)n * a.C);
    const float gs = a.as[n * a.V + v];
    float val = x[k] * gs * a.at[n * a.T + t] * a.ac[n * a.C + c];
    if (a.bwd) val += a.dmv[r * a.T + t] * gs + a.dms[r * a.V + v];
    o[k] = val;
    if (++v == a.V) {
      v = 0;
      if (++t == a.T) { t = 0; ++r; }
    }
  }
  *reinterpret_cast<f32x4*>(a.out + i4) = o;
  if (a.amax) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ok = o[k];
      mx = max(mx, __float_as_uint(ok) & 0x7fffffffu);
    }
  }
  }
  if (a.amax) {                          // one atomic per workgroup at most, none once the running maximum has passed it
    __shared__ unsigned wmax[4];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, k));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
      if (m > __hip_atomic_load(a.amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.amax, m);
    }
  }
}

}  // namespace

static size_t agcn_stc_row_reduce_smem(int T, int V) { return (size_t)(T * V + 32 + T + 8 * 32) * 4; }

extern "C" {

// out_t (N*C, T) = scale_t * sum_v wv[n][v] y*g ; out_v (N*C, V) = scale_v * sum_t wt[n or row][t] y*g ; g / wv / wt /
// either output may be NULL (NULL weights = ones).  One pass over y (and g).
int agcn_stc_row_reduce(const float* y, const float* g, const float* wv, const float* wt, int wt_per_row, float* out_t,
                        float* out_v, float scale_t, float scale_v, int N, int C, int T, int V, void* stream) {
  if (!y || (!out_t && !out_v) || N <= 0 || C <= 0 || T <= 0 || V <= 0 || V > 32) return AGCN_ERR_ARG;
  const size_t smem = agcn_stc_row_reduce_smem(T, V);
  if (smem > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  RowArgs a;
  a.y = y; a.g = g; a.wv = wv; a.wt = wt; a.out_t = out_t; a.out_v = out_v;
  a.scale_t = scale_t; a.scale_v = scale_v; a.N = N; a.C = C; a.T = T; a.V = V; a.wt_per_row = wt_per_row;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(stc_row_reduce_kernel), lds_ok)) return e;
  hipLaunchKernelGGL(stc_row_reduce_kernel, dim3((unsigned)(N * C)), dim3(256), smem, (hipStream_t)stream, a);
  return agcn_check_launch();
}

// out = y * a_s[n,v] * a_t[n,t] * a_c[n,c]      (a_* = 1 + sigmoid gate); N*C*T*V % 4 == 0
int agcn_stc_apply_ex(const float* y, const float* a_s, const float* a_t, const float* a_c, float* out, float* absmax_out,
                      int N, int C, int T, int V, void* stream);
int agcn_stc_apply(const float* y, const float* a_s, const float* a_t, const float* a_c, float* out, int N, int C, int T,
                   int V, void* stream) {
  return agcn_stc_apply_ex(y, a_s, a_t, a_c, out, nullptr, N, C, T, V, stream);
}
// absmax_out (optional, 4 bytes): receives max |out| for the f16x3 temporal convolution that reads the gated tensor next
int agcn_stc_apply_ex(const float* y, const float* a_s, const float* a_t, const float* a_c, float* out, float* absmax_out,
                      int N, int C, int T, int V, void* stream) {
  if (!y || !a_s || !a_t || !a_c || !out || N <= 0 || C <= 0 || T <= 0 || V <= 0) return AGCN_ERR_ARG;
  const long total = (long)N * C * T * V;
  if (total % 4) return AGCN_ERR_UNSUPPORTED;
  if (absmax_out && hipMemsetAsync(absmax_out, 0, 4, (hipStream_t)stream) != hipSuccess) return AGCN_ERR_ARG;
  ApplyArgs a;
  a.y = y; a.as = a_s; a.at = a_t; a.ac = a_c; a.dmv = nullptr; a.dms = nullptr; a.out = out; a.total = total;
  a.C = C; a.T = T; a.V = V; a.bwd = 0; a.amax = reinterpret_cast<unsigned*>(absmax_out);
  hipLaunchKernelGGL(stc_apply_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return agcn_check_launch();
}

// dy = dout * a_s a_t a_c + dmv[n,c,t] * a_s[n,v] + dms[n,c,v]
int agcn_stc_bwd_apply(const float* dout, const float* a_s, const float* a_t, const float* a_c, const float* dmv,
                       const float* dms, float* dy, int N, int C, int T, int V, void* stream) {
  if (!dout || !a_s || !a_t || !a_c || !dmv || !dms || !dy || N <= 0 || C <= 0 || T <= 0 || V <= 0) return AGCN_ERR_ARG;
  const long total = (long)N * C * T * V;
  if (total % 4) return AGCN_ERR_UNSUPPORTED;
  ApplyArgs a;
  a.y = dout; a.as = a_s; a.at = a_t; a.ac = a_c; a.dmv = dmv; a.dms = dms; a.out = dy; a.total = total;
  a.C = C; a.T = T; a.V = V; a.bwd = 1; a.amax = nullptr;
  hipLaunchKernelGGL(stc_apply_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return agcn_check_launch();
}

}  // extern "C"
