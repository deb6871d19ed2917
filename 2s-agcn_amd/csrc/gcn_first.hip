// unit_gcn forward of the FIRST layer (C = 3 input channels, reference agcn.py:92-109 with in_channels = 3):
//     ypre = bias + sum_i Wd_i (x . A^_i)          (aggregate + project, K = 3 subsets x C channels = 9)
//     dpre = bdown + Wdown x                        (the 1x1 `down` convolution, K = C)
// plus the per-tile (sum, sumsq) partials of both for the two BatchNorms that follow.
//
// With 3 channels there is no matrix-core work to speak of (12 multiply-adds per output element): the layer is bound by
// the 2 x (N, 64, T, V) fp32 it has to leave in HBM for the BatchNorm passes (AI = 0.3 FLOP/B).  The generic
// kernels spent 0.31 ms on it (a 32-channel MFMA block of which 3 channels are real, twice, each with its own pass over
// x); this one takes 0.19-0.20 ms: it reads x once, keeps the tile's aggregated 9 x positions in registers and streams both outputs out
// straight from registers.  One wave per workgroup; a lane owns 4 positions of a tile of up to 256 (frame, joint)
// positions, so a row's store is 64 consecutive floats; the BatchNorm partials are summed per lane over its positions
// and combined across lanes once per tile through a small LDS buffer (fixed order).
#include "agcn_common.h"

namespace {

constexpr int L1_NT = 64;     // one wave per workgroup
constexpr int L1_PP = 4;      // consecutive positions per lane (tile = up to 256 positions)
constexpr int L1_M = 32;      // output rows per pass (accumulator registers: 2 * L1_M sums)

struct L1Args {
  const float* x;      // (N, C, T, V)
  const float* adj;    // (N, 3, V, V)
  const float* wcat;   // (M, 3C)
  const float* bias;   // (M) or null
  const float* wdown;  // (M, C) or null
  const float* bdown;  // (M) or null
  float* ypre;         // (N, M, T, V)
  float* ystats;       // [N*ntiles][2][M] or null
  float* dpre;         // (N, M, T, V) (wdown != null)
  float* dstats;       // [N*ntiles][2][M] or null
  int N, M, T, V, ft, ntiles;
};

// Rows m0 .. m0+L1_M-1 of one output tensor for this lane's L1_PP positions: out[o][p] = b[o] + sum_k w[o][k] * g[p][k],
// stored straight from registers (a wave's 64 lanes = 64 consecutive positions of a row: 256-byte runs), and the
// (sum, sumsq) of the tile: per-lane partials over its positions, combined across the 64 lanes through a small LDS
// buffer in a fixed order.
// wl: this tensor's weights in LDS, [M][K+1] with the bias in column K (one broadcast ds_read each: the scalar-load
// version waited a full L2 round trip per output row)
template <int K>
__device__ __forceinline__ void l1_emit(const float* wl, const float (&g)[L1_PP][K], float* __restrict__ out,
                                        float* __restrict__ stats, float* red, int M, int m0, int npos, long row0, long P,
                                        long slot, int lane) {
  float s[L1_M], ss[L1_M];
#pragma unroll
  for (int o = 0; o < L1_M; ++o) {
    const int m = min(m0 + o, M - 1);
    const float bv = wl[m * (K + 1) + K];
    float wk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wk[k] = wl[m * (K + 1) + k];
    float sa = 0.f, sq = 0.f;
    f32x4 vv;
#pragma unroll
    for (int j = 0; j < L1_PP; ++j) {
      float v = bv;
#pragma unroll
      for (int k = 0; k < K; ++k) v += wk[k] * g[j][k];
      vv[j] = v;
    }
    // lane <-> positions 4*lane .. 4*lane+3: a wave stores 1 KB of a row per instruction
    float* dst = out + row0 + (long)(m0 + o) * P + 4 * lane;
    if (4 * lane + 3 < npos) {
      typedef f32x4 f32x4_u __attribute__((aligned(4)));
      if (m0 + o < M) *reinterpret_cast<f32x4_u*>(dst) = vv;
#pragma unroll
      for (int j = 0; j < L1_PP; ++j) { sa += vv[j]; sq += vv[j] * vv[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < L1_PP; ++j)
        if (4 * lane + j < npos) {
          if (m0 + o < M) dst[j] = vv[j];
          sa += vv[j];
          sq += vv[j] * vv[j];
        }
    }
    s[o] = sa; ss[o] = sq;
  }
  if (stats) {
    // cross-lane sums, 32 values at a time: red[lane][33]; lane (half hb, column c) adds 32 rows, the halves combine
    const int c = lane & 31, hb = lane >> 5;
#pragma unroll
    for (int part = 0; part < 2 * L1_M / 32; ++part) {
      const bool sq = part >= L1_M / 32;
      const int o0 = (part % (L1_M / 32)) * 32;
#pragma unroll
      for (int i = 0; i < 32; ++i) red[lane * 33 + i] = sq ? ss[o0 + i] : s[o0 + i];
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): single-wave workgroup, no barrier needed
      float acc = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) acc += red[(hb * 32 + r) * 33 + c];
      acc += __shfl_xor(acc, 32);
      if (hb == 0 && m0 + o0 + c < M) stats[(slot * 2 + (sq ? 1 : 0)) * M + m0 + o0 + c] = acc;
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
  }
}

template <int C>
__global__ void __launch_bounds__(L1_NT) gcn_first_fwd_kernel(const L1Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int V = a.V, ft = a.ft;
  float* xs = smem;                                    // [C][ft*V]
  float* adjs = xs + C * ft * V;                       // [3][V][V]
  float* red = adjs + 3 * V * V;                       // [64][33]
  float* wy = red + 64 * 33;                           // [M][3C+1]
  float* wdl = wy + a.M * (3 * C + 1);                 // [M][C+1]
  const int lane = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);    // neighbouring tiles of a sample share an XCD's L2
  const int n = bid / a.ntiles, tile_id = bid - n * a.ntiles;
  const int t0 = tile_id * ft;
  const int npos = min(ft, a.T - t0) * V;              // valid positions of this tile (<= 64 * L1_PP)
  const long P = (long)a.T * V;
  for (int c = 0; c < C; ++c)
    for (int q = lane; q < npos; q += L1_NT) xs[c * ft * V + q] = a.x[((long)n * C + c) * P + (long)t0 * V + q];
  for (int e = lane; e < 3 * V * V; e += L1_NT) adjs[e] = a.adj[(long)n * 3 * V * V + e];
  for (int e = lane; e < a.M * (3 * C + 1); e += L1_NT) {
    const int m = e / (3 * C + 1), k = e - m * (3 * C + 1);
    wy[e] = k < 3 * C ? a.wcat[m * 3 * C + k] : (a.bias ? a.bias[m] : 0.f);
  }
  if (a.wdown)
    for (int e = lane; e < a.M * (C + 1); e += L1_NT) {
      const int m = e / (C + 1), k = e - m * (C + 1);
      wdl[e] = k < C ? a.wdown[m * C + k] : (a.bdown ? a.bdown[m] : 0.f);
    }
  __syncthreads();
  float g[L1_PP][3 * C], xv[L1_PP][C];
#pragma unroll
  for (int j = 0; j < L1_PP; ++j) {
    const int p = min(4 * lane + j, npos - 1);         // (lanes beyond the tile recompute its last position: unused)
    const int f = p / V, v = p - f * V;
#pragma unroll
    for (int k = 0; k < 3 * C; ++k) g[j][k] = 0.f;
    for (int u = 0; u < V; ++u) {
      float xu[C];
#pragma unroll
      for (int c = 0; c < C; ++c) xu[c] = xs[c * ft * V + f * V + u];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float av = adjs[(i * V + u) * V + v];
#pragma unroll
        for (int c = 0; c < C; ++c) g[j][i * C + c] += xu[c] * av;
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) xv[j][c] = xs[c * ft * V + p];
  }
  const long row0 = (long)n * a.M * P + (long)t0 * V;
  const long slot = (long)n * a.ntiles + tile_id;
  for (int m0 = 0; m0 < a.M; m0 += L1_M) {
    l1_emit<3 * C>(wy, g, a.ypre, a.ystats, red, a.M, m0, npos, row0, P, slot, lane);
    if (a.wdown) l1_emit<C>(wdl, xv, a.dpre, a.dstats, red, a.M, m0, npos, row0, P, slot, lane);
  }
}

inline int l1_frames(int T, int V) {
  int ft = (L1_NT * L1_PP) / V;
  if (ft > T) ft = T;
  return ft < 1 ? 1 : ft;
}

}  // namespace

extern "C" {

// first-layer forward: supported for 1 <= C <= 4 input channels, V <= 32
int agcn_gcn_first_supported(int C, int Cout, int V) { return C >= 1 && C <= 4 && Cout >= 1 && V >= 1 && V <= 32; }

// frame tiles per sample = slots per sample of the two (sum, sumsq) slabs
int agcn_gcn_first_tiles(int T, int V) {
  const int ft = l1_frames(T, V);
  return (T + ft - 1) / ft;
}

int agcn_gcn_first_fwd(const float* x, const float* adj, const float* wcat, const float* bias, const float* wdown,
                       const float* bdown, float* ypre, float* ystats, float* dpre, float* dstats, int N, int C, int Cout,
                       int T, int V, void* stream) {
  if (!x || !adj || !wcat || !ypre || N <= 0 || T <= 0) return AGCN_ERR_ARG;
  if (!agcn_gcn_first_supported(C, Cout, V)) return AGCN_ERR_UNSUPPORTED;
  if (wdown && !dpre) return AGCN_ERR_ARG;
  L1Args a;
  a.x = x; a.adj = adj; a.wcat = wcat; a.bias = bias; a.wdown = wdown; a.bdown = bdown;
  a.ypre = ypre; a.ystats = ystats; a.dpre = dpre; a.dstats = wdown ? dstats : nullptr;
  a.N = N; a.M = Cout; a.T = T; a.V = V;
  a.ft = l1_frames(T, V);
  a.ntiles = (T + a.ft - 1) / a.ft;
  const size_t smem = ((size_t)C * a.ft * V + 3 * V * V + 64 * 33 + (size_t)Cout * (3 * C + 1) + (size_t)Cout * (C + 1)) * 4;
  if (smem > 64 * 1024) return AGCN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(N * a.ntiles));
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 1: hipLaunchKernelGGL(gcn_first_fwd_kernel<1>, grid, dim3(L1_NT), smem, s, a); break;
    case 2: hipLaunchKernelGGL(gcn_first_fwd_kernel<2>, grid, dim3(L1_NT), smem, s, a); break;
    case 3: hipLaunchKernelGGL(gcn_first_fwd_kernel<3>, grid, dim3(L1_NT), smem, s, a); break;
    default: hipLaunchKernelGGL(gcn_first_fwd_kernel<4>, grid, dim3(L1_NT), smem, s, a); break;
  }
  AGCN_NOTE_KERNEL("gcn_first_fwd_kernel<%d>", C);
  return agcn_check_launch();
}

}  // extern "C"
