// Weight gradients of the channel contractions (unit_tcn conv, 1x1 convs, unit_gcn projection), exact fp32 on
// the matrix cores:   dW[m][c][tap] = sum_{n, q} dy[n][m][q] * B(n, c, tap, q)
// where B is the forward B operand (the shifted input window, or the aggregated x . A^_i for the
// projection weights conv_d).  GEMM view: M = output channels, N = (c, tap), K = positions (n, t, v).
//
// A workgroup (8 waves) owns one (m-block, c-block) tile of dW and a contiguous share of the (n, frame-tile)
// pairs (split-K).  Per pair it stages the dy tile and the input window in LDS; a wave owns one 32x32 (m, c)
// tile for a group of taps (the A fragment is reused across the taps), the two wave halves of a SIMD pair split
// the 9 taps 5/4 (or, for 1-tap problems, the positions).  Software pipeline: the global loads of pair p+1 are
// issued before the matrix-core loop of pair p and committed to LDS after it.  Every wave group writes its own
// partial slab and `wgrad_reduce_kernel` sums the slabs in a fixed order (bitwise reproducible; no atomics).
#include "agcn_common.h"

namespace {

struct WgradArgs {
  const float* dy;
  const float* in;
  const float* adj;
  float* part;
  int N, M, C, V, T_src, T_out, stride;
  int tt, ntiles, FW, WLP, DAP, GP;
  long so_m, so_t, so_c;
  long wsize;
  int nsplit, pairs_per_split;
  int off_bx, off_bg, off_adj, off_qoff;
};

// TAPS: 1 or 9 taps (plain) ; AGG: B = x . adj_i with the 3 subsets playing the role of taps.
// 8 waves = MW (m tiles) x CW (c tiles) x TH ; TH=2 splits the taps (KSPLIT=false) or the positions (KSPLIT=true).
// WBX = 64-float column blocks of a window row (bound of the prefetch registers).
template <int TAPS, int AGG, int MW, int CW, int TH, bool KSPLIT, int WBX>
__global__ void __launch_bounds__(512) conv_wgrad_kernel(const WgradArgs a) {
  constexpr int NW = 8, NT = 512;
  static_assert(MW * CW * TH == NW, "8 waves");
  constexpr int BM = MW * 32, CB = CW * 32;
  constexpr int NSUB = AGG ? 3 : 1;
  constexpr int PAD = (TAPS - 1) / 2;
  constexpr bool TSPLIT = (TH == 2) && !KSPLIT;
  constexpr int NTW = TSPLIT ? (TAPS + 1) / 2 : TAPS;     // taps per wave
  constexpr int NACC = NSUB * NTW;
  constexpr int RD = BM / NW, RB = CB / NW;               // dy / window rows per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Da = smem;
  float* Bx = smem + a.off_bx;
  float* Bg = smem + a.off_bg;
  float* adjp = smem + a.off_adj;
  int* qoff = reinterpret_cast<int*>(smem + a.off_qoff);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int th = wave / (MW * CW);
  const int wmc = wave - th * (MW * CW);
  const int mw = wmc % MW, cw = wmc / MW;
  const int ncb = (a.C + CB - 1) / CB;
  const int mb = blockIdx.x / ncb, cb = blockIdx.x - mb * ncb;
  const int m0 = mb * BM, c0 = cb * CB;
  const int V = a.V, tt = a.tt, ttv = tt * V;
  const int Psrc = a.T_src * V, Pout = a.T_out * V;
  const int WL = a.FW * V, WLP = a.WLP, DAP = a.DAP, GP = a.GP;
  const int KS = (ttv + 1) >> 1;
  const int tap0 = TSPLIT ? th * NTW : 0;
  const int ntaps = TSPLIT ? (th == 0 ? NTW : TAPS - NTW) : TAPS;
  const int s_begin = KSPLIT ? (th * ((KS + 1) >> 1)) : 0;
  const int s_end = KSPLIT ? min(KS, s_begin + ((KS + 1) >> 1)) : KS;

  for (int q = tid; q < 2 * KS + 2; q += NT) {
    int o = 0;
    if (q < ttv) {
      const int tl = q / V;
      o = AGG ? q : (tl * a.stride * V + (q - tl * V));
    }
    qoff[q] = o;
  }

  f32x16 acc[NACC];
#pragma unroll
  for (int z = 0; z < NACC; ++z)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[z][j] = 0.f;

  float rda[RD][2];
  float rbx[RB][WBX];

  auto pair_geom = [&](int p, int& n, int& t0, int& nvalid, int& g0) __attribute__((always_inline)) {
    n = p / a.ntiles;
    const int tile = p - n * a.ntiles;
    t0 = tile * tt;
    nvalid = min(tt, a.T_out - t0) * V;
    g0 = (AGG ? t0 : (t0 * a.stride - PAD)) * V;
  };
  // raw loads into registers (predicates are re-evaluated at commit time, so nothing waits on them early)
  auto issue_loads = [&](int p) __attribute__((always_inline)) {
    int n, t0, nvalid, g0;
    pair_geom(p, n, t0, nvalid, g0);
#pragma unroll
    for (int j = 0; j < RD; ++j) {
      const int m = m0 + wave + j * NW;
      const bool okr = m < a.M;
      const float* src = a.dy + ((long)n * a.M + (okr ? m : 0)) * Pout + (long)t0 * V;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = lane + 64 * u;
        rda[j][u] = src[(okr && q < nvalid) ? q : 0];
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int c = c0 + wave + j * NW;
      const bool okr = c < a.C;
      const float* src = a.in + ((long)n * a.C + (okr ? c : 0)) * Psrc;
#pragma unroll
      for (int u = 0; u < WBX; ++u) {
        const int r = lane + 64 * u;
        const int gp = g0 + r;
        rbx[j][u] = src[(okr && r < WL && gp >= 0 && gp < Psrc) ? gp : 0];
      }
    }
  };
  auto commit_lds = [&](int p) __attribute__((always_inline)) {
    int n, t0, nvalid, g0;
    pair_geom(p, n, t0, nvalid, g0);
#pragma unroll
    for (int j = 0; j < RD; ++j) {
      const int ml = wave + j * NW;
      const bool okr = (m0 + ml) < a.M;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = lane + 64 * u;
        if (q < DAP) Da[ml * DAP + q] = (okr && q < nvalid) ? rda[j][u] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int cl = wave + j * NW;
      const bool okr = (c0 + cl) < a.C;
#pragma unroll
      for (int u = 0; u < WBX; ++u) {
        const int r = lane + 64 * u;
        const int gp = g0 + r;
        if (r < WLP) Bx[cl * WLP + r] = (okr && r < WL && gp >= 0 && gp < Psrc) ? rbx[j][u] : 0.f;
      }
    }
  };

  const int total_pairs = a.N * a.ntiles;
  const int p_begin = blockIdx.y * a.pairs_per_split;
  const int p_end = min(total_pairs, p_begin + a.pairs_per_split);
  int last_n = -1;
  if (p_begin < p_end) issue_loads(p_begin);
  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();                          // all waves are done with the previous pair's LDS tiles
    commit_lds(p);
    const int n = p / a.ntiles;
    if (AGG && n != last_n) {
      const int VP = 2 * ((V + 1) / 2);
      const float* adjn = a.adj + (long)n * 3 * V * V;
      for (int e = tid; e < 3 * VP * 32; e += NT) {
        const int i = e / (VP * 32), r = e - i * (VP * 32);
        const int u = r >> 5, col = r & 31;
        const bool ok = u < V && col < V;
        const float t = adjn[ok ? ((i * V + u) * V + col) : 0];
        adjp[e] = ok ? t : 0.f;
      }
      last_n = n;
    }
    if (p + 1 < p_end) issue_loads(p + 1);    // in flight during this pair's matrix-core loop
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (AGG) {
        // Bg[c_local][q] = sum_u Bx[c_local][t*V+u] * adj_sub[u][v]
        if (sub > 0) __syncthreads();
        const int nrows = CB * tt;
        const int nrt = (nrows + 31) >> 5;
        const int VS = (V + 1) >> 1, VP = 2 * VS;
        for (int rt = wave; rt < nrt; rt += NW) {
          const int row = min(rt * 32 + lr, nrows - 1);
          f32x16 d;
#pragma unroll
          for (int j = 0; j < 16; ++j) d[j] = 0.f;
          for (int s = 0; s < VS; ++s) {
            const int u = 2 * s + h;
            float av = Bx[row * V + min(u, V - 1)];
            av = (u < V) ? av : 0.f;
            d = mfma32(av, adjp[(sub * VP + u) * 32 + lr], d);
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int r2 = rt * 32 + mfma_row(j, h);
            if (r2 < nrows && lr < V) {
              const int cl = r2 / tt, tl = r2 - cl * tt;
              Bg[cl * GP + tl * V + lr] = d[j];
            }
          }
        }
        __syncthreads();
      }
      const float* Bsrc = (AGG ? Bg : Bx) + (cw * 32 + lr) * (AGG ? GP : WLP) + tap0 * V;
      const float* Asrc = Da + (mw * 32 + lr) * DAP;
      // operands are fetched one k-step ahead of the MFMAs that use them; for stride 1 the window offset of
      // position q is q itself (no table lookup on the critical path)
      const bool lin = AGG || a.stride == 1;
      auto load_ops = [&](int s, float& av, float (&bv)[NTW]) __attribute__((always_inline)) {
        const int q = 2 * s + h;
        const int qo = lin ? q : qoff[q];
        av = Asrc[q];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bv[t] = Bsrc[qo + ((TSPLIT && t >= ntaps) ? 0 : t * V)];
      };
      if (s_begin < s_end) {
        float av, bv[NTW];
        load_ops(s_begin, av, bv);
        for (int s = s_begin; s < s_end; ++s) {
          float an, bn[NTW];
          load_ops(min(s + 1, s_end - 1), an, bn);
#pragma unroll
          for (int t = 0; t < NTW; ++t)
            if (!TSPLIT || t < ntaps) acc[sub * NTW + t] = mfma32(av, bv[t], acc[sub * NTW + t]);
          av = an;
#pragma unroll
          for (int t = 0; t < NTW; ++t) bv[t] = bn[t];
        }
      }
    }
  }
  // ---- write this wave group's partial slab as [z][m][c] (lanes = consecutive c: coalesced 128-byte rows; the
  //      reduce kernel transposes into the weight layout) ----
  const int slab = KSPLIT ? ((int)blockIdx.y * TH + th) : (int)blockIdx.y;
  float* dst = a.part + (long)slab * a.wsize;
#pragma unroll
  for (int z = 0; z < NACC; ++z) {
    const int sub = z / NTW, t = z - sub * NTW;
    if (TSPLIT && t >= ntaps) continue;
    const int zz = AGG ? sub : (tap0 + t);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int m = m0 + mw * 32 + mfma_row(j, h);
      const int c = c0 + cw * 32 + lr;
      if (m < a.M && c < a.C) dst[((long)zz * a.M + m) * a.C + c] = acc[z][j];
    }
  }
}

// dw[m*so_m + z*so_t + c*so_c] = sum_k part[k][z][m][c]   (fixed order)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long wsize, int nsplit,
                                    int M, int C, long so_m, long so_t, long so_c) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= wsize) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += part[(long)k * wsize + i];
  const int c = (int)(i % C);
  const long r = i / C;
  const int m = (int)(r % M);
  const int z = (int)(r / M);
  dw[(long)m * so_m + (long)z * so_t + (long)c * so_c] = s;
}

// Same sum for many slabs of a small gradient (split-K over up to 1024 workgroup shares): 32 elements x 8 slab
// groups per block; a thread adds every 8th slab (unrolled: 8 loads in flight), the 8 partial sums are combined in
// a fixed order, so the result is still bitwise reproducible.
__global__ void __launch_bounds__(256)
wgrad_reduce_wide_kernel(const float* __restrict__ part, float* __restrict__ dw, long wsize, int nsplit, int M, int C,
                         long so_m, long so_t, long so_c) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + el;
  float s = 0.f;
  if (i < wsize) {
#pragma unroll 8
    for (int k = g; k < nsplit; k += 8) s += part[(long)k * wsize + i];
  }
  red[g][el] = s;
  __syncthreads();
  if (g == 0 && i < wsize) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][el];
    const int c = (int)(i % C);
    const long r = i / C;
    const int m = (int)(r % M);
    const int z = (int)(r / M);
    dw[(long)m * so_m + (long)z * so_t + (long)c * so_c] = t;
  }
}

// picks the reduction shape by slab count
inline int launch_reduce(const float* ws, float* dw, long wsize, int nslabs, int M, int C, long so_m, long so_t, long so_c,
                         hipStream_t stream) {
  if (nslabs >= 64) {
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((wsize + 31) / 32)), dim3(256), 0, stream, ws, dw,
                       wsize, nslabs, M, C, so_m, so_t, so_c);
  } else {
    const int threads = 256;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((wsize + threads - 1) / threads)), dim3(threads), 0,
                       stream, ws, dw, wsize, nslabs, M, C, so_m, so_t, so_c);
  }
  return agcn_check_launch();
}

struct WGeom {
  int tt, ntiles, FW, WLP, DAP, GP, nsplit, nslabs, pairs_per_split, grid_x;
  int off_bx, off_bg, off_adj, off_qoff;
  size_t smem_bytes;
};

template <int TAPS, int AGG, int MW, int CW, int TH, bool KSPLIT>
WGeom wgeom(int N, int M, int C, int V, int T_out, int stride) {
  constexpr int BM = MW * 32, CB = CW * 32;
  WGeom g;
  g.tt = 128 / V;
  if (g.tt > T_out) g.tt = T_out;
  const int ttv = g.tt * V;
  g.ntiles = (T_out + g.tt - 1) / g.tt;
  g.FW = AGG ? g.tt : ((g.tt - 1) * stride + TAPS);
  const int WL = g.FW * V;
  g.WLP = AGG ? WL : (WL | 1);          // odd pitch: lanes index rows (channels) -> conflict-free
  if (!AGG && g.WLP == WL) g.WLP = WL + 2;
  g.DAP = (ttv + 2) | 1;
  g.GP = ttv | 1;
  const int da = BM * g.DAP;
  g.off_bx = (da + 3) & ~3;
  const int bx = CB * g.WLP + 64;
  g.off_bg = g.off_bx + ((bx + 3) & ~3);
  const int bg = AGG ? CB * g.GP + 64 : 0;
  g.off_adj = g.off_bg + ((bg + 3) & ~3);
  const int VP = 2 * ((V + 1) / 2);
  const int adjsz = AGG ? 3 * VP * 32 : 0;
  g.off_qoff = g.off_adj + ((adjsz + 3) & ~3);
  g.smem_bytes = 4 * ((size_t)g.off_qoff + ttv + 8);
  const int nmb = (M + BM - 1) / BM, ncb = (C + CB - 1) / CB;
  g.grid_x = nmb * ncb;
  const int pairs = N * g.ntiles;
  int want = 256 / g.grid_x;            // one 8-wave workgroup per CU
  if (want < 1) want = 1;
  if (want > pairs) want = pairs;
  g.pairs_per_split = (pairs + want - 1) / want;
  g.nsplit = (pairs + g.pairs_per_split - 1) / g.pairs_per_split;
  g.nslabs = g.nsplit * (KSPLIT ? TH : 1);
  return g;
}

template <int TAPS, int AGG, int MW, int CW, int TH, bool KSPLIT, int WBX>
int launch_wgrad(WgradArgs a, float* dw, void* ws, size_t ws_bytes, hipStream_t stream) {
  const WGeom g = wgeom<TAPS, AGG, MW, CW, TH, KSPLIT>(a.N, a.M, a.C, a.V, a.T_out, a.stride);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if (g.WLP > WBX * 64 || g.tt * a.V > 128) return AGCN_ERR_UNSUPPORTED;
  if ((size_t)g.nslabs * a.wsize * 4 > ws_bytes) return AGCN_ERR_WORKSPACE;
  a.part = (float*)ws;
  a.tt = g.tt; a.ntiles = g.ntiles; a.FW = g.FW; a.WLP = g.WLP; a.DAP = g.DAP; a.GP = g.GP;
  a.nsplit = g.nsplit; a.pairs_per_split = g.pairs_per_split;
  a.off_bx = g.off_bx; a.off_bg = g.off_bg; a.off_adj = g.off_adj; a.off_qoff = g.off_qoff;
  auto kern = conv_wgrad_kernel<TAPS, AGG, MW, CW, TH, KSPLIT, WBX>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};   // per (kernel instantiation, device): the attribute is per device
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  hipLaunchKernelGGL(kern, dim3(g.grid_x, g.nsplit), dim3(512), g.smem_bytes, stream, a);
  int rc = agcn_check_launch();
  if (rc) return rc;
  return launch_reduce((const float*)ws, dw, a.wsize, g.nslabs, a.M, a.C, a.so_m, a.so_t, a.so_c, stream);
}

template <int TAPS, int AGG, int MW, int CW, int TH, bool KSPLIT>
size_t ws_wgrad(int N, int M, int C, int V, int T_out, int stride, long wsize) {
  const WGeom g = wgeom<TAPS, AGG, MW, CW, TH, KSPLIT>(N, M, C, V, T_out, stride);
  return (size_t)g.nslabs * wsize * 4;
}

// split-bf16 path (wgrad_chain.hip) + the same fixed-order slab reduction
int chain_wgrad_and_reduce(int agg, const WgradArgs& a, float* dw, void* ws, size_t ws_bytes, hipStream_t stream,
                           const float* dy_absmax = nullptr, const float* x_absmax = nullptr) {
  int nslabs = 0;
  int rc = agcn_wgrad_chain(agg, a.dy, a.in, a.adj, ws, ws_bytes, &nslabs, a.N, a.M, a.C, a.V, a.T_src, a.T_out,
                            a.stride, stream, dy_absmax, x_absmax);
  if (rc) return rc;
  return launch_reduce((const float*)ws, dw, a.wsize, nslabs, a.M, a.C, a.so_m, a.so_t, a.so_c, stream);
}

// 9-tap weight gradient on the split-bf16 kernel (AGCN_WGRAD9_BF16=0 keeps the exact-f32 MFMA kernel)
inline bool wgrad9_bf16_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AGCN_WGRAD9_BF16");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v == 1;
}

inline bool chain_wgrad_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AGCN_WGRAD_BF16");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v == 1 && agcn_chained();
}

}  // namespace

extern "C" {

// bytes of workspace agcn_conv_bwd_weight / agcn_gcn_project_bwd_weight need
size_t agcn_conv_bwd_weight_workspace(int N, int Cin, int Cout, int T, int V, int taps, int stride) {
  const int pad = (taps - 1) / 2;
  const int T_out = (T + 2 * pad - taps) / stride + 1;
  const long wsize = (long)Cout * Cin * taps;
  if (taps == 9) {
    size_t b9 = (Cout % 128 == 0) ? ws_wgrad<9, 0, 4, 1, 2, false>(N, Cout, Cin, V, T_out, stride, wsize)
                                  : ws_wgrad<9, 0, 2, 2, 2, false>(N, Cout, Cin, V, T_out, stride, wsize);
    if (agcn_wgrad9_bf16_supported(Cout, Cin, V, stride)) {
      const size_t t = agcn_wgrad9_bf16_workspace(N, Cout, Cin, V, T, stride);
      if (t > b9) b9 = t;
    }
    return b9;
  }
  size_t b = (Cout % 128 == 0) ? ws_wgrad<1, 0, 4, 2, 1, false>(N, Cout, Cin, V, T_out, stride, wsize)
                              : ws_wgrad<1, 0, 2, 2, 2, true>(N, Cout, Cin, V, T_out, stride, wsize);
  if (agcn_wgrad_chain_supported(Cout, Cin, V)) {
    const size_t t = agcn_wgrad_chain_workspace(0, N, Cout, Cin, V, T_out);
    if (t > b) b = t;
  }
  return b;
}

// dw[o][c][k] = sum_{n,t,v} dy[n][o][t,v] * x[n][c][(t*stride + k - pad), v]
int agcn_conv_bwd_weight_ex(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes, int N,
                            int Cin, int Cout, int T, int V, int taps, int stride, const float* dy_absmax,
                            const float* x_absmax, void* stream);
int agcn_conv_bwd_weight(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes, int N,
                         int Cin, int Cout, int T, int V, int taps, int stride, void* stream) {
  return agcn_conv_bwd_weight_ex(dy, x, dw, workspace, workspace_bytes, N, Cin, Cout, T, V, taps, stride, nullptr, nullptr,
                                 stream);
}
// dy_absmax / x_absmax: device scalars max |dy| / max |x| their producers left behind; with BOTH given the tap-free
// gradient (taps = 1, stride 1, Cin a multiple of 64) runs on f16x3, otherwise on bf16x6 as before (no pass of its own)
int agcn_conv_bwd_weight_ex(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes, int N,
                            int Cin, int Cout, int T, int V, int taps, int stride, const float* dy_absmax,
                            const float* x_absmax, void* stream) {
  if (!dy || !x || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if ((taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return AGCN_ERR_UNSUPPORTED;
  const int pad = (taps - 1) / 2;
  WgradArgs a = {};
  a.dy = dy; a.in = x; a.N = N; a.M = Cout; a.C = Cin; a.V = V; a.T_src = T;
  a.T_out = (T + 2 * pad - taps) / stride + 1; a.stride = stride;
  a.so_m = (long)Cin * taps; a.so_t = 1; a.so_c = taps; a.wsize = (long)Cout * Cin * taps;
  hipStream_t s = (hipStream_t)stream;
  if (taps == 9) {
    if (wgrad9_bf16_enabled() && agcn_wgrad9_bf16_supported(Cout, Cin, V, stride)) {
      int nslabs = 0;
      int rc = agcn_wgrad9_bf16(dy, x, workspace, workspace_bytes, &nslabs, N, Cout, Cin, V, T, stride, s, dy_absmax, x_absmax);
      if (rc) return rc;
      return launch_reduce((const float*)workspace, dw, a.wsize, nslabs, a.M, a.C, a.so_m, a.so_t, a.so_c, s);
    }
    if (stride == 1) {
      if (Cout % 128 == 0) return launch_wgrad<9, 0, 4, 1, 2, false, 6>(a, dw, workspace, workspace_bytes, s);
      return launch_wgrad<9, 0, 2, 2, 2, false, 6>(a, dw, workspace, workspace_bytes, s);
    }
    if (Cout % 128 == 0) return launch_wgrad<9, 0, 4, 1, 2, false, 7>(a, dw, workspace, workspace_bytes, s);
    return launch_wgrad<9, 0, 2, 2, 2, false, 7>(a, dw, workspace, workspace_bytes, s);
  }
  if (chain_wgrad_enabled() && agcn_wgrad_chain_supported(Cout, Cin, V))
    return chain_wgrad_and_reduce(0, a, dw, workspace, workspace_bytes, s, dy_absmax, x_absmax);
  if (Cout % 128 == 0) return launch_wgrad<1, 0, 4, 2, 1, false, 4>(a, dw, workspace, workspace_bytes, s);
  return launch_wgrad<1, 0, 2, 2, 2, true, 4>(a, dw, workspace, workspace_bytes, s);
}

size_t agcn_gcn_project_bwd_weight_workspace(int N, int C, int Cout, int T, int V) {
  const long wsize = 3L * Cout * C;
  size_t b = (Cout % 128 == 0) ? ws_wgrad<1, 1, 4, 2, 1, false>(N, Cout, C, V, T, 1, wsize)
                              : ws_wgrad<1, 1, 2, 2, 2, true>(N, Cout, C, V, T, 1, wsize);
  if (agcn_wgrad_chain_supported(Cout, C, V)) {
    const size_t t = agcn_wgrad_chain_workspace(1, N, Cout, C, V, T);
    if (t > b) b = t;
  }
  return b;
}

// dwcat[o][i*C+c] = sum_{n,t,v} dy[n][o][t,v] * sum_u x[n][c][t,u] adj[n][i][u][v]
int agcn_gcn_project_bwd_weight_ex(const float* dy, const float* x, const float* adj, float* dwcat, void* workspace,
                                   size_t workspace_bytes, int N, int C, int Cout, int T, int V, const float* dy_absmax,
                                   const float* x_absmax, void* stream);
int agcn_gcn_project_bwd_weight(const float* dy, const float* x, const float* adj, float* dwcat, void* workspace,
                                size_t workspace_bytes, int N, int C, int Cout, int T, int V, void* stream) {
  return agcn_gcn_project_bwd_weight_ex(dy, x, adj, dwcat, workspace, workspace_bytes, N, C, Cout, T, V, nullptr, nullptr,
                                        stream);
}
// with BOTH maxima given (and C a multiple of 64) the contraction runs on f16x3, otherwise on bf16x6 as before
int agcn_gcn_project_bwd_weight_ex(const float* dy, const float* x, const float* adj, float* dwcat, void* workspace,
                                   size_t workspace_bytes, int N, int C, int Cout, int T, int V, const float* dy_absmax,
                                   const float* x_absmax, void* stream) {
  if (!dy || !x || !adj || !dwcat || !workspace || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  WgradArgs a = {};
  a.dy = dy; a.in = x; a.adj = adj; a.N = N; a.M = Cout; a.C = C; a.V = V; a.T_src = T; a.T_out = T; a.stride = 1;
  a.so_m = 3L * C; a.so_t = C; a.so_c = 1; a.wsize = 3L * Cout * C;
  hipStream_t s = (hipStream_t)stream;
  if (chain_wgrad_enabled() && agcn_wgrad_chain_supported(Cout, C, V))
    return chain_wgrad_and_reduce(1, a, dwcat, workspace, workspace_bytes, s, dy_absmax, x_absmax);
  if (Cout % 128 == 0) return launch_wgrad<1, 1, 4, 2, 1, false, 2>(a, dwcat, workspace, workspace_bytes, s);
  return launch_wgrad<1, 1, 2, 2, 2, true, 2>(a, dwcat, workspace, workspace_bytes, s);
}

}  // extern "C"
