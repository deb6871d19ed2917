// Weight gradients of the channel contractions (unit_tcn conv, 1x1 convs, unit_gcn projection), exact fp32 on
// the matrix cores:   dW[m][c][tap] = sum_{n, q} dy[n][m][q] * B(n, c, tap, q)
// where B is the forward B operand (the shifted input window, or the aggregated x . A^_i for the
// projection weights conv_d).  GEMM view: M = output channels, N = (c, tap), K = positions (n, t, v).
// Each workgroup owns one (m-block, c-block) tile of dW and a contiguous share of the (n, frame-tile)
// pairs (split-K); it writes one partial slab, and `wgrad_reduce_kernel` sums the slabs in a fixed order
// (bitwise reproducible; no float atomics).
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

// TAPS: 1 or 9 taps (plain) ; AGG: B = x . adj_i with the 3 subsets playing the role of taps
template <int TAPS, int AGG, int MW, int CW, int TMr, int TNr>
__global__ void __launch_bounds__(MW* CW * 64) conv_wgrad_kernel(const WgradArgs a) {
  constexpr int NW = MW * CW, NT = NW * 64;
  constexpr int BM = MW * TMr * 32, CB = CW * TNr * 32;
  static_assert((BM / NW) % 8 == 0 && (CB / NW) % 8 == 0, "staging batches rows by 8");
  constexpr int NSUB = AGG ? 3 : 1;
  constexpr int PAD = (TAPS - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Da = smem;
  float* Bx = smem + a.off_bx;
  float* Bg = smem + a.off_bg;
  float* adjp = smem + a.off_adj;
  int* qoff = reinterpret_cast<int*>(smem + a.off_qoff);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int mw = wave % MW, cw = wave / MW;
  const int ncb = (a.C + CB - 1) / CB;
  const int mb = blockIdx.x / ncb, cb = blockIdx.x - mb * ncb;
  const int m0 = mb * BM, c0 = cb * CB;
  const int V = a.V, tt = a.tt, ttv = tt * V;
  const int Psrc = a.T_src * V, Pout = a.T_out * V;
  const int WL = a.FW * V, WLP = a.WLP, DAP = a.DAP, GP = a.GP;
  const int KS = (ttv + 1) >> 1;

  for (int q = tid; q < 2 * KS + 2; q += NT) {
    int o = 0;
    if (q < ttv) {
      const int tl = q / V;
      o = AGG ? q : (tl * a.stride * V + (q - tl * V));
    }
    qoff[q] = o;
  }

  f32x16 acc[TMr][TNr][NSUB * TAPS];
#pragma unroll
  for (int x = 0; x < TMr; ++x)
#pragma unroll
    for (int y = 0; y < TNr; ++y)
#pragma unroll
      for (int z = 0; z < NSUB * TAPS; ++z)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[x][y][z][j] = 0.f;

  const int total_pairs = a.N * a.ntiles;
  const int p_begin = blockIdx.y * a.pairs_per_split;
  const int p_end = min(total_pairs, p_begin + a.pairs_per_split);
  int last_n = -1;
  for (int p = p_begin; p < p_end; ++p) {
    const int n = p / a.ntiles, tile = p - n * a.ntiles;
    const int t0 = tile * tt;
    const int nvalid = min(tt, a.T_out - t0) * V;
    const int f0 = AGG ? t0 : (t0 * a.stride - PAD);
    __syncthreads();
    // ---- stage dy tile: Da[m_local][q] (zero beyond the valid positions, incl. the pad column).
    //      Loads are issued in batches (8 rows x column blocks) before the LDS stores: many requests in flight. ----
    {
      constexpr int RPW = BM / NW;        // rows per wave (16 or 32)
      constexpr int RG = 8;               // rows per batch
      const long dybase = (long)n * a.M * Pout + (long)t0 * V;
      for (int jg = 0; jg < RPW; jg += RG) {
        for (int q0 = lane; q0 < DAP; q0 += 128) {
          float val[RG][2];
#pragma unroll
          for (int j = 0; j < RG; ++j) {
            const int ml = wave + (jg + j) * NW;
            const int m = m0 + ml;
            const bool okr = m < a.M;
            const float* src = a.dy + dybase + (long)(okr ? m : 0) * Pout;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int q = q0 + 64 * u;
              const bool ok = okr && q < nvalid;
              const float t = src[ok ? q : 0];
              val[j][u] = ok ? t : 0.f;
            }
          }
#pragma unroll
          for (int j = 0; j < RG; ++j) {
            const int ml = wave + (jg + j) * NW;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int q = q0 + 64 * u;
              if (q < DAP) Da[ml * DAP + q] = val[j][u];
            }
          }
        }
      }
    }
    // ---- stage input window: Bx[c_local][r] ----
    {
      constexpr int RPW = CB / NW;        // 8, 16 or 32
      constexpr int RG = 8;
      const int g0 = f0 * V;
      for (int jg = 0; jg < RPW; jg += RG) {
        for (int r0 = lane; r0 < WLP; r0 += 128) {
          float val[RG][2];
#pragma unroll
          for (int j = 0; j < RG; ++j) {
            const int cl = wave + (jg + j) * NW;
            const int c = c0 + cl;
            const bool okr = c < a.C;
            const float* src = a.in + ((long)n * a.C + (okr ? c : 0)) * Psrc;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int r = r0 + 64 * u;
              const int gp = g0 + r;
              const bool ok = okr && r < WL && gp >= 0 && gp < Psrc;
              const float t = src[ok ? gp : 0];
              val[j][u] = ok ? t : 0.f;
            }
          }
#pragma unroll
          for (int j = 0; j < RG; ++j) {
            const int cl = wave + (jg + j) * NW;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int r = r0 + 64 * u;
              if (r < WLP) Bx[cl * WLP + r] = val[j][u];
            }
          }
        }
      }
    }
    if (AGG && n != last_n) {
      const int VP = 2 * ((V + 1) / 2);
      const float* adjn = a.adj + (long)n * 3 * V * V;
      for (int e = tid; e < 3 * VP * 32; e += NT) {
        const int i = e / (VP * 32), r = e - i * (VP * 32);
        const int u = r >> 5, col = r & 31;
        adjp[e] = (u < V && col < V) ? adjn[(i * V + u) * V + col] : 0.f;
      }
      last_n = n;
    }
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
      const float* Bsrc = AGG ? Bg : Bx;
      const int BP = AGG ? GP : WLP;
      for (int s = 0; s < KS; ++s) {
        const int q = 2 * s + h;
        const int qo = qoff[q];
        float av[TMr], bv[TNr][TAPS];
#pragma unroll
        for (int x = 0; x < TMr; ++x) av[x] = Da[((mw * TMr + x) * 32 + lr) * DAP + q];
#pragma unroll
        for (int y = 0; y < TNr; ++y)
#pragma unroll
          for (int tap = 0; tap < TAPS; ++tap) bv[y][tap] = Bsrc[((cw * TNr + y) * 32 + lr) * BP + qo + tap * V];
#pragma unroll
        for (int x = 0; x < TMr; ++x)
#pragma unroll
          for (int y = 0; y < TNr; ++y)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
              acc[x][y][sub * TAPS + tap] = mfma32(av[x], bv[y][tap], acc[x][y][sub * TAPS + tap]);
      }
    }
  }
  // ---- write the partial slab in the final weight layout ----
  float* dst = a.part + (long)blockIdx.y * a.wsize;
#pragma unroll
  for (int x = 0; x < TMr; ++x)
#pragma unroll
    for (int y = 0; y < TNr; ++y)
#pragma unroll
      for (int z = 0; z < NSUB * TAPS; ++z)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int m = m0 + (mw * TMr + x) * 32 + mfma_row(j, h);
          const int c = c0 + (cw * TNr + y) * 32 + lr;
          if (m < a.M && c < a.C) dst[(long)m * a.so_m + (long)z * a.so_t + (long)c * a.so_c] = acc[x][y][z][j];
        }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long wsize, int nsplit) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= wsize) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += part[(long)k * wsize + i];
  dw[i] = s;
}

struct WGeom {
  int tt, ntiles, FW, WLP, DAP, GP, nsplit, pairs_per_split, grid_x;
  int off_bx, off_bg, off_adj, off_qoff;
  size_t smem_bytes;
};

template <int TAPS, int AGG, int MW, int CW, int TMr, int TNr>
WGeom wgeom(int N, int M, int C, int V, int T_out, int stride) {
  constexpr int BM = MW * TMr * 32, CB = CW * TNr * 32;
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
  int want = 512 / g.grid_x;
  if (want < 1) want = 1;
  if (want > pairs) want = pairs;
  g.pairs_per_split = (pairs + want - 1) / want;
  g.nsplit = (pairs + g.pairs_per_split - 1) / g.pairs_per_split;
  return g;
}

template <int TAPS, int AGG, int MW, int CW, int TMr, int TNr>
int launch_wgrad(WgradArgs a, float* dw, void* ws, size_t ws_bytes, hipStream_t stream) {
  const WGeom g = wgeom<TAPS, AGG, MW, CW, TMr, TNr>(a.N, a.M, a.C, a.V, a.T_out, a.stride);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if ((size_t)g.nsplit * a.wsize * 4 > ws_bytes) return AGCN_ERR_WORKSPACE;
  a.part = (float*)ws;
  a.tt = g.tt; a.ntiles = g.ntiles; a.FW = g.FW; a.WLP = g.WLP; a.DAP = g.DAP; a.GP = g.GP;
  a.nsplit = g.nsplit; a.pairs_per_split = g.pairs_per_split;
  a.off_bx = g.off_bx; a.off_bg = g.off_bg; a.off_adj = g.off_adj; a.off_qoff = g.off_qoff;
  auto kern = conv_wgrad_kernel<TAPS, AGG, MW, CW, TMr, TNr>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.grid_x, g.nsplit), dim3(MW * CW * 64), g.smem_bytes, stream, a);
  int rc = agcn_check_launch();
  if (rc) return rc;
  const int threads = 256;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((a.wsize + threads - 1) / threads)), dim3(threads), 0,
                     stream, (const float*)ws, dw, a.wsize, g.nsplit);
  return agcn_check_launch();
}

template <int TAPS, int AGG, int MW, int CW, int TMr, int TNr>
size_t ws_wgrad(int N, int M, int C, int V, int T_out, int stride, long wsize) {
  const WGeom g = wgeom<TAPS, AGG, MW, CW, TMr, TNr>(N, M, C, V, T_out, stride);
  return (size_t)g.nsplit * wsize * 4;
}

}  // namespace

extern "C" {

// bytes of workspace agcn_conv_bwd_weight / agcn_gcn_project_bwd_weight need (upper bound over configs)
size_t agcn_conv_bwd_weight_workspace(int N, int Cin, int Cout, int T, int V, int taps, int stride) {
  const int pad = (taps - 1) / 2;
  const int T_out = (T + 2 * pad - taps) / stride + 1;
  const long wsize = (long)Cout * Cin * taps;
  if (taps == 9) {
    if (Cout % 128 == 0) return ws_wgrad<9, 0, 4, 1, 1, 1>(N, Cout, Cin, V, T_out, stride, wsize);
    return ws_wgrad<9, 0, 2, 2, 1, 1>(N, Cout, Cin, V, T_out, stride, wsize);
  }
  if (Cout % 128 == 0) return ws_wgrad<1, 0, 4, 1, 1, 2>(N, Cout, Cin, V, T_out, stride, wsize);
  return ws_wgrad<1, 0, 2, 2, 1, 1>(N, Cout, Cin, V, T_out, stride, wsize);
}

// dw[o][c][k] = sum_{n,t,v} dy[n][o][t,v] * x[n][c][(t*stride + k - pad), v]
int agcn_conv_bwd_weight(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes, int N,
                         int Cin, int Cout, int T, int V, int taps, int stride, void* stream) {
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
    if (Cout % 128 == 0) return launch_wgrad<9, 0, 4, 1, 1, 1>(a, dw, workspace, workspace_bytes, s);
    return launch_wgrad<9, 0, 2, 2, 1, 1>(a, dw, workspace, workspace_bytes, s);
  }
  if (Cout % 128 == 0) return launch_wgrad<1, 0, 4, 1, 1, 2>(a, dw, workspace, workspace_bytes, s);
  return launch_wgrad<1, 0, 2, 2, 1, 1>(a, dw, workspace, workspace_bytes, s);
}

size_t agcn_gcn_project_bwd_weight_workspace(int N, int C, int Cout, int T, int V) {
  const long wsize = 3L * Cout * C;
  if (Cout % 128 == 0) return ws_wgrad<1, 1, 4, 1, 1, 2>(N, Cout, C, V, T, 1, wsize);
  return ws_wgrad<1, 1, 2, 2, 1, 1>(N, Cout, C, V, T, 1, wsize);
}

// dwcat[o][i*C+c] = sum_{n,t,v} dy[n][o][t,v] * sum_u x[n][c][t,u] adj[n][i][u][v]
int agcn_gcn_project_bwd_weight(const float* dy, const float* x, const float* adj, float* dwcat, void* workspace,
                                size_t workspace_bytes, int N, int C, int Cout, int T, int V, void* stream) {
  if (!dy || !x || !adj || !dwcat || !workspace || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  WgradArgs a = {};
  a.dy = dy; a.in = x; a.adj = adj; a.N = N; a.M = Cout; a.C = C; a.V = V; a.T_src = T; a.T_out = T; a.stride = 1;
  a.so_m = 3L * C; a.so_t = C; a.so_c = 1; a.wsize = 3L * Cout * C;
  hipStream_t s = (hipStream_t)stream;
  if (Cout % 128 == 0) return launch_wgrad<1, 1, 4, 1, 1, 2>(a, dwcat, workspace, workspace_bytes, s);
  return launch_wgrad<1, 1, 2, 2, 1, 1>(a, dwcat, workspace, workspace_bytes, s);
}

}  // extern "C"
