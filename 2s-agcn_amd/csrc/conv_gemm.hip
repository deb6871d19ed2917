// Implicit-GEMM channel contraction on NCHW skeleton tensors, exact fp32 on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32).  One kernel template covers
//   * unit_tcn's (9x1)/(1x1) temporal convolution forward           (reference agcn.py:40-41,49)
//   * its backward-data (transposed, tap-flipped; stride 1 and 2)
//   * every 1x1 conv of unit_gcn (conv_a/conv_b/down)               (reference agcn.py:66-75,99-100)
//   * unit_gcn's fused aggregate+project  y = sum_i Wd_i (x . A^_i)  (reference agcn.py:103-105)
//     and its backward-data  dx = sum_i Wd_i^T (dy . A^_i^T)
//   * the adjacency gradient  dA^_i[u,v] = sum_{c,t} x[c,t,u] (Wd_i^T dy)[c,t,v]
//
// GEMM view per sample n and per tile of `tt` whole frames (tt*V <= BN positions):
//   out[n][m][q] = sum_{kc,tap} A(m,kc,tap) * B(n,kc,tap,q),   q = t_local*V + v
// A (weights) and the B source window are staged in LDS per K-chunk of CK channels; every one of
// the TAPS taps re-reads the SAME staged window at an offset of tap*V floats, so the 9x reuse of the
// temporal convolution is served from LDS, not HBM.  For the aggregated variants the B chunk is first
// multiplied by the (per-sample) VxV adjacency, also on the matrix cores, and kept in LDS.
#include "agcn_common.h"

namespace {

struct ConvGemmArgs {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  const float* adj;    // AGG: (N,3,V,V)
  float* stats;        // [N*ntiles][2][M] partial (sum, sum of squares) or null
  const float* add1;
  const float* mask1;
  const float* add2;
  const float* mask2;
  const float* xin;    // DADJ: x (N,C,P)
  float* dadj;         // DADJ: partial buffer
  int N, M, Kinner, in_rows;
  int V, T_src, T_out, stride, tt, ntiles;
  int FW;              // frames staged per window row
  int WLP;             // Bx row pitch (floats)
  long sa_m, sa_i, sa_c;
  int accumulate;
  int C;               // DADJ: channels of x
  int off_bx, off_bg, off_adj;   // LDS offsets (floats)
};

__device__ __forceinline__ int floordiv2(int x) { return x >> 1; }   // arithmetic shift = floor

template <int MODE, int TAPS, int AGG, bool S2, int WM, int WN, int TM, int TN, int CK, int EPI>
__global__ void __launch_bounds__(WM* WN * 64) conv_gemm_kernel(const ConvGemmArgs a) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int BM = WM * TM * 32, BMP = BM + 1;
  constexpr int NSUB = AGG ? 3 : 1;
  constexpr int PAD = (TAPS - 1) / 2;
  constexpr int KK = NSUB * CK * TAPS;   // rows of the staged A chunk
  static_assert(CK % 2 == 0, "CK must be even");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Aw = smem;
  float* Bx = smem + a.off_bx;
  float* Bg = smem + a.off_bg;
  float* adjp = smem + a.off_adj;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int n = bid / a.ntiles, tile = bid - n * a.ntiles;
  const int m0 = blockIdx.y * BM;
  const int V = a.V, tt = a.tt, t0 = tile * tt;
  const int ttv = tt * V;
  const int tvalid = min(tt, a.T_out - t0);
  const int nvalid = tvalid * V;
  const int Psrc = a.T_src * V;
  const int Pout = a.T_out * V;

  int f0;
  if (AGG) f0 = t0;
  else if (MODE == 0) f0 = t0 * a.stride - PAD;
  else if (!S2) f0 = t0 - PAD;
  else f0 = floordiv2(t0 - PAD);
  const int hb0 = S2 ? (t0 - PAD - 2 * f0) : 0;
  const int WL = a.FW * V;
  const int WLP = a.WLP;

  // per-lane B offsets for the TN position tiles of this wave
  int boff[TN], vq[TN], hbq[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int q = (wn * TN + tn) * 32 + lr;
    if (q >= ttv) q = 0;              // padding lanes read position 0; results are discarded
    const int tl = q / V, v = q - tl * V;
    vq[tn] = v;
    hbq[tn] = hb0 + tl;
    if (AGG) boff[tn] = q;
    else if (MODE == 0) boff[tn] = tl * a.stride * V + v;
    else boff[tn] = q;
  }

  if (AGG) {
    // zero-padded adjacency fragments: adjp[i][u][col], u < VP (even), col < 32
    const int VP = 2 * ((V + 1) / 2);
    const float* adjn = a.adj + (long)n * 3 * V * V;
    for (int e = tid; e < 3 * VP * 32; e += NT) {
      const int i = e / (VP * 32), r = e - i * (VP * 32);
      const int u = r >> 5, col = r & 31;
      float val = 0.f;
      if (u < V && col < V) val = (AGG == 1) ? adjn[(i * V + u) * V + col] : adjn[(i * V + col) * V + u];
      adjp[e] = val;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[tm][tn][j] = 0.f;

  const int nchunks = (a.Kinner + CK - 1) / CK;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int kc0 = ch * CK;
    __syncthreads();   // every wave is done reading the previous chunk
    // ---- stage A chunk: Aw[kk][m_local]; loads are issued in batches of AU before any LDS store so that
    //      AU global requests are in flight per lane (a load->store->load chain is latency-bound) ----
    {
      constexpr int AU = 6;
      constexpr int TOTAL = BM * KK;
      for (int e0 = tid; e0 < TOTAL; e0 += NT * AU) {
        float val[AU];
        int dst[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
          const int e = min(e0 + u * NT, TOTAL - 1);
          int ml, i, kcl, gt, row;
          if (MODE == 0) {
            ml = e / KK;
            const int kk = e - ml * KK;
            i = kk / (CK * TAPS);
            const int r = kk - i * (CK * TAPS);
            kcl = r / TAPS;
            gt = r - kcl * TAPS;
            row = kk;
          } else {
            const int kkc = e / (BM * TAPS), r = e - kkc * (BM * TAPS);
            ml = r / TAPS;
            gt = r - ml * TAPS;
            i = kkc / CK;
            kcl = kkc - i * CK;
            row = kkc * TAPS + (TAPS - 1 - gt);
          }
          const int m = m0 + ml, kc = kc0 + kcl;
          const bool ok = m < a.M && kc < a.Kinner;
          const long gi = ok ? ((long)m * a.sa_m + (long)i * a.sa_i + (long)kc * a.sa_c + gt) : 0;
          const float t = a.w[gi];
          val[u] = ok ? t : 0.f;
          dst[u] = row * BMP + ml;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u)
          if (e0 + u * NT < TOTAL) Aw[dst[u]] = val[u];
      }
    }
    // ---- stage B source window: Bx[kc_local][r], r = (f - f0)*V + v, zero outside [0,T_src) ----
    {
      constexpr int RPW = (CK + NW - 1) / NW;            // rows per wave
      constexpr int RU = (RPW >= 8) ? 1 : (8 / RPW);     // column blocks in flight per row
      const float* rowp[RPW];
      bool rok[RPW];
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const int kcl = wave + j * NW;
        rok[j] = kcl < CK && (kc0 + kcl) < a.Kinner;
        rowp[j] = a.in + ((long)n * a.in_rows + (rok[j] ? (kc0 + kcl) : 0)) * Psrc;
      }
      const int g0 = f0 * V;
      for (int r0 = lane; r0 < WL; r0 += 64 * RU) {
        float val[RPW][RU];
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
          for (int u = 0; u < RU; ++u) {
            const int r = r0 + 64 * u;
            const int gp = g0 + r;
            const bool ok = rok[j] && r < WL && gp >= 0 && gp < Psrc;
            const float t = rowp[j][ok ? gp : 0];
            val[j][u] = ok ? t : 0.f;
          }
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
          for (int u = 0; u < RU; ++u) {
            const int r = r0 + 64 * u;
            const int kcl = wave + j * NW;
            if (kcl < CK && r < WL) Bx[kcl * WLP + r] = val[j][u];
          }
      }
    }
    __syncthreads();
    if (AGG) {
      // Bg[i][c_local][q] = sum_u Bx[c_local][t*V+u] * adj_i[u][v]   (rows r=(c_local,t), 32 per MFMA tile)
      const int nrows = CK * tt;
      const int nrt = (nrows + 31) >> 5;
      const int VS = (V + 1) >> 1;
      for (int tl = wave; tl < 3 * nrt; tl += NW) {
        const int i = tl / nrt, rt = tl - i * nrt;
        const int row = min(rt * 32 + lr, nrows - 1);
        f32x16 d;
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = 0.f;
        const int VP = 2 * VS;
        for (int s = 0; s < VS; ++s) {
          const int u = 2 * s + h;
          float av = Bx[row * V + min(u, V - 1)];
          av = (u < V) ? av : 0.f;
          const float bv = adjp[(i * VP + u) * 32 + lr];
          d = mfma32(av, bv, d);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int r2 = rt * 32 + mfma_row(j, h);
          if (r2 < nrows && lr < V) Bg[i * CK * ttv + r2 * V + lr] = d[j];
        }
      }
      __syncthreads();
    }
    // ---- matrix-core contraction over this chunk ----
    constexpr int KP = (NSUB * CK) / 2;
    for (int kp = 0; kp < KP; ++kp) {
      const int krow = 2 * kp + h;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        float av[TM], bv[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) av[tm] = Aw[(krow * TAPS + tap) * BMP + (wm * TM + tm) * 32 + lr];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if (AGG) {
            bv[tn] = Bg[krow * ttv + boff[tn]];
          } else if (!S2) {
            bv[tn] = Bx[krow * WLP + boff[tn] + tap * V];
          } else {
            const int num = hbq[tn] + tap;
            const float val = Bx[krow * WLP + (num >> 1) * V + vq[tn]];
            bv[tn] = (num & 1) ? 0.f : val;
          }
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(av[tm], bv[tn], acc[tm][tn]);
      }
    }
  }

  if (EPI == 0) {
    // ---- store (+bias, +accumulate, +masked addends) and per-channel (sum, sumsq) partials ----
    float* red = smem;   // aliases Aw: [WN][2][BM]
    if (a.stats) __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int ml = (wm * TM + tm) * 32 + mfma_row(j, h);
        const int m = m0 + ml;
        const bool mok = m < a.M;
        const float bval = (a.bias && mok) ? a.bias[m] : 0.f;
        float bsum = 0.f, bsq = 0.f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int q = (wn * TN + tn) * 32 + lr;
          if (mok && q < nvalid) {
            const long idx = ((long)n * a.M + m) * Pout + (long)t0 * V + q;
            float val = acc[tm][tn][j] + bval;
            if (a.accumulate) val += a.out[idx];
            if (a.add1) {
              float t = a.add1[idx];
              if (a.mask1) t = (a.mask1[idx] > 0.f) ? t : 0.f;
              val += t;
            }
            if (a.add2) {
              float t = a.add2[idx];
              if (a.mask2) t = (a.mask2[idx] > 0.f) ? t : 0.f;
              val += t;
            }
            a.out[idx] = val;
            bsum += val;
            bsq += val * val;
          }
        }
        if (a.stats) {
          bsum = half_sum(bsum);
          bsq = half_sum(bsq);
          if (lr == 0) {
            red[(wn * 2 + 0) * BM + ml] = bsum;
            red[(wn * 2 + 1) * BM + ml] = bsq;
          }
        }
      }
    }
    if (a.stats) {
      __syncthreads();
      const long slot = (long)n * a.ntiles + tile;
      for (int e = tid; e < 2 * BM; e += NT) {
        const int k = e / BM, ml = e - k * BM;
        float s = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WN; ++w2) s += red[(w2 * 2 + k) * BM + ml];
        if (m0 + ml < a.M) a.stats[(slot * 2 + k) * a.M + m0 + ml] = s;
      }
    }
  } else {
    // ---- DADJ: dadj_i[u][v] = sum_{c,t} x[c][t,u] * acc[(i,c)][t,v]; rows of this block are m=(i,c) ----
    const int C = a.C;
    float* Dg = smem;                 // [BM][ttv]
    float* Xs = smem + BM * ttv;      // [BM][ttv]
    float* red2 = Xs + BM * ttv;      // [NW][V*V]
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int ml = (wm * TM + tm) * 32 + mfma_row(j, h);
          const int q = (wn * TN + tn) * 32 + lr;
          if (q < ttv) Dg[ml * ttv + q] = (q < nvalid && m0 + ml < a.M) ? acc[tm][tn][j] : 0.f;
        }
    for (int ml = wave; ml < BM; ml += NW) {
      const int m = m0 + ml;
      const bool ok = m < a.M;
      const int c = ok ? (m % C) : 0;
      const float* src = a.xin + ((long)n * C + c) * Pout + (long)t0 * V;
      for (int q = lane; q < ttv; q += 64) Xs[ml * ttv + q] = (ok && q < nvalid) ? src[q] : 0.f;
    }
    __syncthreads();
    const int nmb = (C >= BM) ? (C / BM) : 1;
    const int VV = V * V;
    for (int i = 0; i < 3; ++i) {
      const int c_lo = max(0, m0 - i * C), c_hi = min(C, m0 + BM - i * C);
      if (c_lo >= c_hi) continue;   // block-uniform
      const int npairs = (c_hi - c_lo + 1) >> 1;
      f32x16 d;
#pragma unroll
      for (int j = 0; j < 16; ++j) d[j] = 0.f;
      const int lc = min(lr, V - 1);
      for (int it = wave; it < npairs * tt; it += NW) {
        const int ap = it / tt, tl = it - ap * tt;
        const int c = c_lo + 2 * ap + h;
        const bool ok = (c < c_hi) && (lr < V);
        const int ml = i * C + min(c, c_hi - 1) - m0;
        float av = Xs[ml * ttv + tl * V + lc];
        float bv = Dg[ml * ttv + tl * V + lc];
        av = ok ? av : 0.f;
        bv = ok ? bv : 0.f;
        d = mfma32(av, bv, d);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int u = mfma_row(j, h);
        if (u < V && lr < V) red2[wave * VV + u * V + lr] = d[j];
      }
      __syncthreads();
      const int slot = tile * nmb + ((C >= BM) ? ((int)blockIdx.y - i * nmb) : 0);
      float* dst = a.dadj + (((long)n * 3 + i) * ((long)a.ntiles * nmb) + slot) * VV;
      for (int e = tid; e < VV; e += NT) {
        float s = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) s += red2[w2 * VV + e];
        dst[e] = s;
      }
      __syncthreads();
    }
  }
}

struct Geometry {
  int tt, ntiles, FW, WLP, ttv;
  size_t smem_bytes;
  int off_bx, off_bg, off_adj;
};

// host-side tile geometry shared by the launcher and the workspace queries
template <int MODE, int TAPS, int AGG, bool S2, int WM, int WN, int TM, int TN, int CK, int EPI>
Geometry make_geometry(int V, int T_out, int stride) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BMP = BM + 1, NSUB = AGG ? 3 : 1, NW = WM * WN;
  Geometry g;
  g.tt = BN / V;
  if (g.tt > T_out) g.tt = T_out;
  g.ttv = g.tt * V;
  g.ntiles = (T_out + g.tt - 1) / g.tt;
  if (AGG) g.FW = g.tt;
  else if (MODE == 0) g.FW = (g.tt - 1) * stride + TAPS;
  else if (!S2) g.FW = g.tt + TAPS - 1;
  else g.FW = (g.tt + TAPS - 1) / 2 + 2;
  g.WLP = AGG ? g.ttv : g.FW * V + 8;
  const int aw = NSUB * CK * TAPS * BMP;
  g.off_bx = (aw + 3) & ~3;
  const int bx = CK * g.WLP + 64;
  g.off_bg = g.off_bx + ((bx + 3) & ~3);
  const int bgsz = AGG ? (3 * CK * g.ttv + 64) : 0;
  g.off_adj = g.off_bg + ((bgsz + 3) & ~3);
  const int VP = 2 * ((V + 1) / 2);
  const int adjsz = AGG ? 3 * VP * 32 : 0;
  size_t main_f = (size_t)g.off_adj + adjsz;
  size_t epi_f = (EPI == 0) ? (size_t)WN * 2 * BM : (size_t)2 * BM * g.ttv + (size_t)NW * V * V;
  g.smem_bytes = 4 * (main_f > epi_f ? main_f : epi_f);
  return g;
}

template <int MODE, int TAPS, int AGG, bool S2, int WM, int WN, int TM, int TN, int CK, int EPI>
int launch_cfg(ConvGemmArgs a, hipStream_t stream) {
  constexpr int BM = WM * TM * 32;
  const Geometry g = make_geometry<MODE, TAPS, AGG, S2, WM, WN, TM, TN, CK, EPI>(a.V, a.T_out, a.stride);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  a.tt = g.tt;
  a.ntiles = g.ntiles;
  a.FW = g.FW;
  a.WLP = g.WLP;
  a.off_bx = g.off_bx;
  a.off_bg = g.off_bg;
  a.off_adj = g.off_adj;
  auto kern = conv_gemm_kernel<MODE, TAPS, AGG, S2, WM, WN, TM, TN, CK, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid((unsigned)(a.N * g.ntiles), (unsigned)((a.M + BM - 1) / BM));
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), g.smem_bytes, stream, a);
  return agcn_check_launch();
}

// BM = 64 (4 waves) unless M is a multiple of 128 (8 waves, BM = 128)
#define DISPATCH_BM(MODE, TAPS, AGG, S2, CK64, CK128, a, s)                                  \
  (((a).M % 128 == 0) ? launch_cfg<MODE, TAPS, AGG, S2, 2, 4, 2, 2, CK128, 0>((a), (s))      \
                      : launch_cfg<MODE, TAPS, AGG, S2, 1, 4, 2, 2, CK64, 0>((a), (s)))

}  // namespace

extern "C" {

// frames per tile / tiles per sample used by the stats-partial layout of the conv kernels
int agcn_conv_tile_frames(int V, int T_out) {
  int tt = 256 / V;
  return tt > T_out ? T_out : tt;
}
int agcn_conv_num_tiles(int V, int T_out) {
  int tt = agcn_conv_tile_frames(V, T_out);
  return (T_out + tt - 1) / tt;
}
int agcn_dadj_num_slots(int C, int V, int T) {
  int tt = 128 / V;
  if (tt > T) tt = T;
  int ntiles = (T + tt - 1) / tt;
  int nmb = C >= 64 ? C / 64 : 1;
  return ntiles * nmb;
}

// y[n][o][t,v] = bias[o] + sum_{c,k} w[o][c][k] x[n][c][(t*stride + k - pad), v]      (unit_tcn conv, 1x1 convs)
int agcn_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* stats_part, int N, int Cin,
                  int Cout, int T, int V, int taps, int stride, void* stream) {
  if (!x || !w || !y || N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32) return AGCN_ERR_ARG;
  if ((taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return AGCN_ERR_UNSUPPORTED;
  const int pad = (taps - 1) / 2;
  ConvGemmArgs a = {};
  a.in = x; a.w = w; a.bias = bias; a.out = y; a.stats = stats_part;
  a.N = N; a.M = Cout; a.Kinner = Cin; a.in_rows = Cin; a.V = V;
  a.T_src = T; a.T_out = (T + 2 * pad - taps) / stride + 1; a.stride = stride;
  a.sa_m = (long)Cin * taps; a.sa_i = 0; a.sa_c = taps;
  hipStream_t s = (hipStream_t)stream;
  if (taps == 9) return DISPATCH_BM(0, 9, 0, false, 16, 8, a, s);
  return DISPATCH_BM(0, 1, 0, false, 32, 32, a, s);
}

// dx[n][c][t,v] (+)= sum_{o,k} w[o][c][k] dy[n][o][(t + pad - k)/stride, v]  (+ masked addends)
int agcn_conv_bwd_data(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                       const float* mask1, const float* add2, const float* mask2, int N, int Cin, int Cout, int T,
                       int V, int taps, int stride, void* stream) {
  if (!dy || !w || !dx || N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32) return AGCN_ERR_ARG;
  if ((taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return AGCN_ERR_UNSUPPORTED;
  const int pad = (taps - 1) / 2;
  ConvGemmArgs a = {};
  a.in = dy; a.w = w; a.out = dx; a.accumulate = accumulate;
  a.add1 = add1; a.mask1 = mask1; a.add2 = add2; a.mask2 = mask2;
  a.N = N; a.M = Cin; a.Kinner = Cout; a.in_rows = Cout; a.V = V;
  a.T_src = (T + 2 * pad - taps) / stride + 1; a.T_out = T; a.stride = stride;
  a.sa_m = taps; a.sa_i = 0; a.sa_c = (long)Cin * taps;
  hipStream_t s = (hipStream_t)stream;
  if (taps == 9) {
    if (stride == 2) return DISPATCH_BM(1, 9, 0, true, 16, 8, a, s);
    return DISPATCH_BM(1, 9, 0, false, 16, 8, a, s);
  }
  if (stride == 2) return DISPATCH_BM(1, 1, 0, true, 32, 32, a, s);
  return DISPATCH_BM(1, 1, 0, false, 32, 32, a, s);
}

// y[n][o][t,v] = bias[o] + sum_i sum_c wcat[o][i*C+c] * sum_u x[n][c][t,u] adj[n][i][u][v]
int agcn_gcn_aggregate_project_fwd(const float* x, const float* adj, const float* wcat, const float* bias, float* y,
                                   float* stats_part, int N, int C, int Cout, int T, int V, void* stream) {
  if (!x || !adj || !wcat || !y || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32) return AGCN_ERR_ARG;
  ConvGemmArgs a = {};
  a.in = x; a.w = wcat; a.bias = bias; a.out = y; a.adj = adj; a.stats = stats_part;
  a.N = N; a.M = Cout; a.Kinner = C; a.in_rows = C; a.V = V; a.T_src = T; a.T_out = T; a.stride = 1;
  a.sa_m = 3L * C; a.sa_i = C; a.sa_c = 1;
  return DISPATCH_BM(0, 1, 1, false, 8, 8, a, (hipStream_t)stream);
}

// dx[n][c][t,u] (+)= sum_i sum_o wcat[o][i*C+c] * sum_v dy[n][o][t,v] adj[n][i][u][v]   (+ masked addends)
int agcn_gcn_aggregate_project_bwd_data(const float* dy, const float* adj, const float* wcat, float* dx,
                                        int accumulate, const float* add1, const float* mask1, const float* add2,
                                        const float* mask2, int N, int C, int Cout, int T, int V, void* stream) {
  if (!dy || !adj || !wcat || !dx || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32) return AGCN_ERR_ARG;
  ConvGemmArgs a = {};
  a.in = dy; a.w = wcat; a.out = dx; a.adj = adj; a.accumulate = accumulate;
  a.add1 = add1; a.mask1 = mask1; a.add2 = add2; a.mask2 = mask2;
  a.N = N; a.M = C; a.Kinner = Cout; a.in_rows = Cout; a.V = V; a.T_src = T; a.T_out = T; a.stride = 1;
  a.sa_m = 1; a.sa_i = C; a.sa_c = 3L * C;
  return DISPATCH_BM(1, 1, 2, false, 8, 8, a, (hipStream_t)stream);
}

// dadj_part[n][i][slot][u][v] = sum over the slot's (c,t) of x[n][c][t,u] * (sum_o wcat[o][i*C+c] dy[n][o][t,v])
int agcn_gcn_dadj(const float* dy, const float* wcat, const float* x, float* dadj_part, int N, int C, int Cout,
                  int T, int V, void* stream) {
  if (!dy || !wcat || !x || !dadj_part || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if (C >= 64 && C % 64 != 0) return AGCN_ERR_UNSUPPORTED;
  ConvGemmArgs a = {};
  a.in = dy; a.w = wcat; a.xin = x; a.dadj = dadj_part;
  a.N = N; a.M = 3 * C; a.Kinner = Cout; a.in_rows = Cout; a.V = V; a.T_src = T; a.T_out = T; a.stride = 1;
  a.sa_m = 1; a.sa_i = 0; a.sa_c = 3L * C; a.C = C;
  return launch_cfg<1, 1, 0, false, 1, 4, 2, 1, 32, 1>(a, (hipStream_t)stream);
}

}  // extern "C"
