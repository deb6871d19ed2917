// Implicit-GEMM channel contraction on NCHW skeleton tensors, exact fp32 on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32).  One kernel template covers
//   * unit_tcn's (9x1)/(1x1) temporal convolution forward           (reference agcn.py:40-41,49)
//   * its backward-data (transposed, tap-flipped; stride 2 split by output-frame parity into a 5-tap and a
//     4-tap stride-1 problem so no matrix-core work is spent on structural zeros)
//   * every 1x1 conv of unit_gcn (conv_a/conv_b/down)               (reference agcn.py:66-75,99-100)
//   * unit_gcn's fused aggregate+project  y = sum_i Wd_i (x . A^_i)  (reference agcn.py:103-105)
//     and its backward-data  dx = sum_i Wd_i^T (dy . A^_i^T)
//   * the adjacency gradient  dA^_i[u,v] = sum_{c,t} x[c,t,u] (Wd_i^T dy)[c,t,v]
//
// GEMM view per sample n and per tile of `tt` whole frames (tt*V <= BN positions):
//   out[n][m][q] = sum_{kc,tap} A(m,kc,tap) * B(n,kc,tap,q),   q = t_local*V + v
// * A: a tiny pack kernel first lays the weights out as the exact LDS image of every (row block, K chunk)
//   (zero padded, tap-flipped / transposed as the mode needs), so the hot kernel stages A with contiguous float4.
// * B: per K-chunk of CK channels the SOURCE WINDOW (tt+taps-1 frames) is staged in LDS once and every tap reads the
//   same window at +tap*V floats: the 9x reuse of the temporal convolution is served from LDS, not HBM.
//   For the aggregated variants the chunk is first multiplied by the sample's padded VxV adjacency on the matrix
//   cores and kept in LDS.
// * Software pipeline (register prefetch): the global loads of chunk k+1 are issued before the matrix-core loop of
//   chunk k and committed to LDS after it, so HBM/L2 latency hides under the MFMAs; operand fragments are read one
//   step ahead of the MFMAs that use them.
#include "agcn_common.h"
#include "split_f16.h"

namespace {

struct ConvGemmArgs {
  const float* in;
  const float* wp;     // packed weight images [mblocks][nchunks][KK][BM]
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
  int V, T_src, T_out, tt, ntiles;   // T_out = number of output frame slots tau this launch covers
  int src_stride, f_off;             // first source frame of a tile: t0*src_stride + f_off
  int out_fs, out_fo, T_full;        // output frame = tau*out_fs + out_fo ; T_full = frames of the out tensor
  int FW, WLP;
  int accumulate;
  int mask_bits;                     // mask1/mask2 are sign bit masks instead of fp32 tensors
  int C;                             // DADJ: channels of x
  int nchunks, nmb;
  int off_bx, off_bg, off_adj;       // LDS offsets (floats)
};

struct PackArgs {
  const float* w;
  float* wp;
  int M, Kinner, nchunks;
  long sa_m, sa_i, sa_c;
  int tap_mul, tap_add, tap_flip_from;   // weight tap = tap_flip_from >= 0 ? tap_flip_from - (j*tap_mul+tap_add) : j
};

// wp[((mb*nchunks + ch)*KK + kk)*BM + ml],  kk = (i*CK + kcl)*TAPS + j
template <int TAPS, int NSUB, int CK, int BM>
__global__ void __launch_bounds__(256) pack_weights_kernel(const PackArgs p) {
  constexpr int KK = NSUB * CK * TAPS;
  const int ch = blockIdx.x % p.nchunks, mb = blockIdx.x / p.nchunks;
  float* dst = p.wp + (long)blockIdx.x * KK * BM;
  for (int e = threadIdx.x; e < KK * BM; e += 256) {
    const int kk = e / BM, ml = e - kk * BM;
    const int i = kk / (CK * TAPS), r = kk - i * (CK * TAPS);
    const int kcl = r / TAPS, j = r - kcl * TAPS;
    const int m = mb * BM + ml, kc = ch * CK + kcl;
    const int gt = p.tap_flip_from >= 0 ? (p.tap_flip_from - (j * p.tap_mul + p.tap_add)) : j;
    float v = 0.f;
    if (m < p.M && kc < p.Kinner) v = p.w[(long)m * p.sa_m + (long)i * p.sa_i + (long)kc * p.sa_c + gt];
    dst[e] = v;
  }
}

// One 32x32 tile of the aggregation G = X . A^: all NS operand pairs are fetched from LDS first, then the NS
// dependent matrix-core steps run back to back (one LDS latency per tile instead of one per step).
// Rows u >= V of the padded adjacency image are zero, so the clamped x operand needs no masking.
template <int S0, int NS, int VMIN>   // VMIN = smallest V this chain is used with: only u >= VMIN needs the clamp
__device__ __forceinline__ f32x16 agg_chain(const float* bxrow, const float* adjcol, int V, int h, f32x16 d) {
  float av[NS], bv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int u = 2 * (S0 + s) + h;
    av[s] = (2 * (S0 + s) + 1 < VMIN) ? bxrow[u] : bxrow[min(u, V - 1)];
    bv[s] = adjcol[u * 32];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) d = mfma32(av[s], bv[s], d);
  __builtin_amdgcn_sched_barrier(0);   // keep the next chain's operand fetches from being hoisted above this one
  return d;
}

// WB = max number of 64-float column blocks of a staged window row (compile-time bound of the prefetch registers)
template <int TAPS, int AGG, int WM, int WN, int TM, int TN, int CK, int WB, int EPI>
__global__ void __launch_bounds__(WM* WN * 64, (TAPS == 1 && EPI == 0 && (AGG == 0 || WM * WN == 8) && (WB < 8 || WM * WN == 8)) ? 4 : 2) conv_gemm_kernel(const ConvGemmArgs a) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int NSUB = AGG ? 3 : 1;
  constexpr int KK = NSUB * CK * TAPS;       // rows of the staged A chunk
  constexpr int N4 = KK * BM / 4;            // float4s of one A image
  constexpr int EA = (N4 + NT - 1) / NT;     // A float4 per thread
  constexpr int RPW = (CK + NW - 1) / NW;    // B rows per wave
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
  // 1-D grid: the row blocks of one (sample, frame tile) are adjacent (they stage the same source window, so
  // it is served from the XCD's L2), and xcd_remap keeps neighbouring tiles on one XCD
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mbk = bid % a.nmb;
  const int nt_id = bid / a.nmb;
  const int n = nt_id / a.ntiles, tile = nt_id - n * a.ntiles;
  const int m0 = mbk * BM;
  const int V = a.V, tt = a.tt, t0 = tile * tt;
  const int ttv = tt * V;
  const int tvalid = min(tt, a.T_out - t0);
  const int nvalid = tvalid * V;
  const int Psrc = a.T_src * V;
  const int f0 = t0 * a.src_stride + a.f_off;
  const int WL = a.FW * V;
  const int WLP = a.WLP;

  // per-lane B offsets / output offsets for the TN position tiles of this wave
  int boff[TN], ooff[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int q = (wn * TN + tn) * 32 + lr;
    if (q >= ttv) q = 0;              // padding lanes read position 0; their results are discarded
    const int tl = q / V, v = q - tl * V;
    boff[tn] = AGG ? q : (tl * a.src_stride * V + v);
    ooff[tn] = ((t0 + tl) * a.out_fs + a.out_fo) * V + v;
  }

  if (AGG) {
    // zero-padded adjacency fragments: adjp[i][u][col], u < VP (even), col < 32
    const int VP = 2 * ((V + 1) / 2);
    const float* adjn = a.adj + (long)n * 3 * V * V;
    for (int e = tid; e < 3 * VP * 32; e += NT) {
      const int i = e / (VP * 32), r = e - i * (VP * 32);
      const int u = r >> 5, col = r & 31;
      const bool ok = u < V && col < V;
      const int gi = ok ? ((AGG == 1) ? ((i * V + u) * V + col) : ((i * V + col) * V + u)) : 0;
      const float t = adjn[gi];
      adjp[e] = ok ? t : 0.f;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[tm][tn][j] = 0.f;

  // ---- prefetch registers ----
  f32x4 ra[EA];
  float rb[RPW][WB];
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(a.wp) + (long)mbk * a.nchunks * N4;
  const int g0 = f0 * V;

  // NOTE: the loaded values are kept RAW in registers; range predicates are re-evaluated at commit time, so
  // nothing consumes a load result before the matrix-core loop it is meant to hide under.
  auto issue_loads = [&](int ch) __attribute__((always_inline)) {
    const f32x4* src = wp4 + (long)ch * N4;
#pragma unroll
    for (int u = 0; u < EA; ++u) ra[u] = src[min(tid + u * NT, N4 - 1)];
    const int kc0 = ch * CK;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int kcl = wave + j * NW;
      const bool rok = kcl < CK && (kc0 + kcl) < a.Kinner;
      const float* rowp = a.in + ((long)n * a.in_rows + (rok ? (kc0 + kcl) : 0)) * Psrc;
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int r = lane + 64 * u;
        const int gp = g0 + r;
        const bool ok = rok && r < WL && gp >= 0 && gp < Psrc;
        rb[j][u] = rowp[ok ? gp : 0];
      }
    }
  };
  auto commit_lds = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < EA; ++u)
      if (tid + u * NT < N4) reinterpret_cast<f32x4*>(Aw)[tid + u * NT] = ra[u];
    const int kc0 = ch * CK;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int kcl = wave + j * NW;
      const bool rok = kcl < CK && (kc0 + kcl) < a.Kinner;
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int r = lane + 64 * u;
        const int gp = g0 + r;
        const bool ok = rok && gp >= 0 && gp < Psrc;
        if (kcl < CK && r < WL) Bx[kcl * WLP + r] = ok ? rb[j][u] : 0.f;
      }
    }
  };

  // DADJ: the x tile of the epilogue is fetched now (raw, into registers) so its latency hides under the K loop
  constexpr int XR = (EPI == 1) ? (BM / NW) : 1;
  float rx[XR][2];
  if (EPI == 1) {
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int m = m0 + wave + j * NW;
      const bool okr = m < a.M;
      const float* src = a.xin + ((long)n * a.C + (okr ? (m % a.C) : 0)) * ((long)a.T_full * V) + (long)t0 * V;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = lane + 64 * u;
        rx[j][u] = src[(okr && q < nvalid) ? q : 0];
      }
    }
  }

  const int nchunks = a.nchunks;
  if (nchunks > 0) issue_loads(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();                       // every wave is done reading the previous chunk from LDS
    commit_lds(ch);
    if (ch + 1 < nchunks) issue_loads(ch + 1);   // in flight during this chunk's matrix-core loop
    __syncthreads();
    if (AGG) {
      // Bg[i][c_local][q] = sum_u Bx[c_local][t*V+u] * adj_i[u][v]   (rows r=(c_local,t), 32 per MFMA tile)
      const int nrows = CK * tt;
      const int nrt = (nrows + 31) >> 5;
      const int VS = (V + 1) >> 1;
      const int VP = 2 * VS;
      for (int tl = wave; tl < 3 * nrt; tl += NW) {
        const int i = tl / nrt, rt = tl - i * nrt;
        const int row = min(rt * 32 + lr, nrows - 1);
        const float* bxrow = Bx + row * V;
        const float* adjcol = adjp + i * VP * 32 + lr;
        f32x16 d;
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = 0.f;
        if (NW != 8 && VS == 13) {          // V = 25 (NTU)
          d = agg_chain<0, 13, 25>(bxrow, adjcol, V, h, d);
        } else if (NW != 8 && VS == 9) {    // V = 18 (Kinetics)
          d = agg_chain<0, 9, 17>(bxrow, adjcol, V, h, d);
        } else {
          // 8-wave variant (128-register budget) and other V: plain loop (a longer unrolled chain spills there)
          for (int s = 0; s < VS; ++s) {
            const int u = 2 * s + h;
            d = mfma32(bxrow[min(u, V - 1)], adjcol[u * 32], d);
          }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int r2 = rt * 32 + mfma_row(j, h);
          if (r2 < nrows && lr < V) Bg[i * CK * ttv + r2 * V + lr] = d[j];
        }
      }
      __syncthreads();
    }
    // ---- matrix-core contraction over this chunk; operands are fetched one step ahead ----
    constexpr int KP = (NSUB * CK) / 2;
    const float* Bsrc = AGG ? Bg : Bx;
    const int BP = AGG ? ttv : WLP;
    const float* Arow = Aw + (wm * TM) * 32 + lr;
    float av[TM], bv[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) av[tm] = Arow[(h * TAPS) * BM + tm * 32];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bv[tn] = Bsrc[h * BP + boff[tn]];
#pragma unroll 1
    for (int kp = 0; kp < KP; ++kp) {
      const int krow = 2 * kp + h;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        // next step (clamped at the end: a harmless re-read)
        const int ntap = (tap == TAPS - 1) ? 0 : tap + 1;
        const int nkrow = (tap == TAPS - 1) ? min(krow + 2, 2 * (KP - 1) + h) : krow;
        float an[TM], bn[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) an[tm] = Arow[(nkrow * TAPS + ntap) * BM + tm * 32];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bn[tn] = Bsrc[nkrow * BP + boff[tn] + (AGG ? 0 : ntap * V)];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(av[tm], bv[tn], acc[tm][tn]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) av[tm] = an[tm];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bv[tn] = bn[tn];
      }
    }
  }

  if (EPI == 0) {
    // ---- store (+bias, +accumulate, +masked addends) and per-channel (sum, sumsq) partials ----
    float* red = smem;   // aliases Aw: [WN][2][BM]
    const long Pfull = (long)a.T_full * V;
    if (a.stats) __syncthreads();
    const bool has_extra = a.accumulate || a.add1 || a.add2;   // kernel-uniform
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        // 4 rows x TN columns per batch: all residual/accumulate loads of the batch are issued together, with
        // clamped (always valid) addresses, instead of one exec-masked load -> wait per element
        float ex[4][TN];
        long idxs[4][TN];
        bool oks[4][TN];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int m = m0 + (wm * TM + tm) * 32 + mfma_row(jb * 4 + jj, h);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const int q = (wn * TN + tn) * 32 + lr;
            oks[jj][tn] = (m < a.M) && (q < nvalid);
            idxs[jj][tn] = oks[jj][tn] ? (((long)n * a.M + m) * Pfull + ooff[tn]) : 0;
            ex[jj][tn] = 0.f;
          }
        }
        if (has_extra) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              const long idx = idxs[jj][tn];
              float e = 0.f;
              if (a.accumulate) e += a.out[idx];
              if (a.add1) {
                float t = a.add1[idx];
                if (a.mask1) t = mask_pass(mask_load(a.mask1, idx, a.mask_bits), idx, a.mask_bits) ? t : 0.f;
                e += t;
              }
              if (a.add2) {
                float t = a.add2[idx];
                if (a.mask2) t = mask_pass(mask_load(a.mask2, idx, a.mask_bits), idx, a.mask_bits) ? t : 0.f;
                e += t;
              }
              ex[jj][tn] = e;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = jb * 4 + jj;
          const int ml = (wm * TM + tm) * 32 + mfma_row(j, h);
          const int m = m0 + ml;
          const float bval = (a.bias && m < a.M) ? a.bias[m] : 0.f;
          float bsum = 0.f, bsq = 0.f;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const float val = acc[tm][tn][j] + bval + ex[jj][tn];
            if (oks[jj][tn]) {
              a.out[idxs[jj][tn]] = val;
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
    const int Pout = a.T_full * V;
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
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int ml = wave + j * NW;
      const bool okr = (m0 + ml) < a.M;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = lane + 64 * u;
        if (q < ttv) Xs[ml * ttv + q] = (okr && q < nvalid) ? rx[j][u] : 0.f;
      }
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
        float xv = Xs[ml * ttv + tl * V + lc];
        float gv = Dg[ml * ttv + tl * V + lc];
        xv = ok ? xv : 0.f;
        gv = ok ? gv : 0.f;
        d = mfma32(xv, gv, d);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int u = mfma_row(j, h);
        if (u < V && lr < V) red2[wave * VV + u * V + lr] = d[j];
      }
      __syncthreads();
      const int slot = tile * nmb + ((C >= BM) ? (mbk - i * nmb) : 0);
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
  int tt, ntiles, FW, WLP, ttv, nchunks, nmb;
  size_t smem_bytes, pack_floats;
  int off_bx, off_bg, off_adj;
};

// host-side tile geometry shared by the launcher and the workspace queries
template <int TAPS, int AGG, int WM, int WN, int TM, int TN, int CK, int EPI>
Geometry make_geometry(int V, int T_out, int src_stride, int M, int Kinner) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NSUB = AGG ? 3 : 1, NW = WM * WN;
  Geometry g;
  g.tt = BN / V;
  if (g.tt > T_out) g.tt = T_out;
  if (g.tt < 1) g.tt = 1;
  g.ttv = g.tt * V;
  g.ntiles = (T_out + g.tt - 1) / g.tt;
  g.FW = AGG ? g.tt : ((g.tt - 1) * src_stride + TAPS);
  g.WLP = AGG ? g.ttv : g.FW * V + 8;
  g.nchunks = (Kinner + CK - 1) / CK;
  g.nmb = (M + BM - 1) / BM;
  const int aw = NSUB * CK * TAPS * BM;
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
  g.pack_floats = (size_t)g.nmb * g.nchunks * aw;
  return g;
}

struct Problem {          // what differs between the entry points
  ConvGemmArgs a;
  const float* w;
  long sa_m, sa_i, sa_c;
  int tap_mul, tap_add, tap_flip_from;
  void* ws;
  size_t ws_bytes;
};

template <int TAPS, int AGG, int WM, int WN, int TM, int TN, int CK, int WB, int EPI>
int launch_cfg(Problem& p, hipStream_t stream) {
  constexpr int BM = WM * TM * 32, NSUB = AGG ? 3 : 1;
  ConvGemmArgs a = p.a;
  const Geometry g = make_geometry<TAPS, AGG, WM, WN, TM, TN, CK, EPI>(a.V, a.T_out, a.src_stride, a.M, a.Kinner);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if (g.FW * a.V > WB * 64) return AGCN_ERR_UNSUPPORTED;
  if (g.pack_floats * 4 > p.ws_bytes) return AGCN_ERR_WORKSPACE;
  a.tt = g.tt; a.ntiles = g.ntiles; a.FW = g.FW; a.WLP = g.WLP; a.nchunks = g.nchunks; a.nmb = g.nmb;
  a.off_bx = g.off_bx; a.off_bg = g.off_bg; a.off_adj = g.off_adj;
  a.wp = (const float*)p.ws;
  if (g.nchunks > 0) {
    PackArgs pk;
    pk.w = p.w; pk.wp = (float*)p.ws; pk.M = a.M; pk.Kinner = a.Kinner; pk.nchunks = g.nchunks;
    pk.sa_m = p.sa_m; pk.sa_i = p.sa_i; pk.sa_c = p.sa_c;
    pk.tap_mul = p.tap_mul; pk.tap_add = p.tap_add; pk.tap_flip_from = p.tap_flip_from;
    hipLaunchKernelGGL((pack_weights_kernel<TAPS, NSUB, CK, BM>), dim3(g.nmb * g.nchunks), dim3(256), 0, stream, pk);
    int rc = agcn_check_launch();
    if (rc) return rc;
  }
  auto kern = conv_gemm_kernel<TAPS, AGG, WM, WN, TM, TN, CK, WB, EPI>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};   // per (kernel instantiation, device): the attribute is per device
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  dim3 grid((unsigned)(a.N * g.ntiles * g.nmb));
  AGCN_NOTE_KERNEL("conv_gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d>", TAPS, AGG, WM, WN, TM, TN, CK, WB, EPI);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), g.smem_bytes, stream, a);
  return agcn_check_launch();
}

template <int TAPS, int AGG, int WM, int WN, int TM, int TN, int CK, int EPI>
size_t pack_bytes(int V, int T_out, int src_stride, int M, int Kinner) {
  return 4 * make_geometry<TAPS, AGG, WM, WN, TM, TN, CK, EPI>(V, T_out, src_stride, M, Kinner).pack_floats;
}

// BM = 64 (4 waves) unless M is a multiple of 128 (8 waves, BM = 128)
#define DISPATCH_BM(TAPS, AGG, CK64, CK128, WB, p, s)                                        \
  (((p).a.M % 128 == 0) ? launch_cfg<TAPS, AGG, 2, 4, 2, 2, CK128, WB, 0>((p), (s))          \
                        : launch_cfg<TAPS, AGG, 1, 4, 2, 2, CK64, WB, 0>((p), (s)))
#define PACK_BYTES_BM(TAPS, AGG, CK64, CK128, V, T, ss, M, K)                                \
  (((M) % 128 == 0) ? pack_bytes<TAPS, AGG, 2, 4, 2, 2, CK128, 0>(V, T, ss, M, K)            \
                    : pack_bytes<TAPS, AGG, 1, 4, 2, 2, CK64, 0>(V, T, ss, M, K))

constexpr int CK9 = 8, CK1 = 16, CKA = 8, CKD = 64;


}  // namespace

extern "C" {

// frames per tile / tiles per sample used by the stats-partial layout of the conv kernels
int agcn_conv_tile_frames(int V, int T_out) {
  int tt = 256 / V;
  return tt > T_out ? T_out : tt;
}
// diagnostic: kernel instantiation the calling thread's last contraction launch used ("" if none)
const char* agcn_last_kernel(void) { return agcn_last_kernel_buf; }
// "bf16x6" | "f32" | "bf16x3": the arithmetic of the channel contractions, fixed per process by AGCN_GEMM
const char* agcn_gemm_mode(void) {
  const int m = agcn_gemm_precision();
  return m == 3 ? "bf16x6" : (m == 0 ? "f32" : (m == 1 ? "bf16" : "bf16x3"));
}
// *out = max |x| over n floats: the streaming pass the f16x3 kernels run themselves when no producer supplied the maximum
int agcn_absmax(const float* x, long n, float* out, void* stream) {
  if (!x || !out || n <= 0) return AGCN_ERR_ARG;
  return agcn_launch_absmax(x, n, reinterpret_cast<unsigned*>(out), (hipStream_t)stream);
}
// arithmetic of unit_gcn's aggregate+project chain (forward and backward-data): "f16x3" in the default fp32-equivalent
// mode (AGCN_CHAIN_F16X3=0: "bf16x6"), else AGCN_GEMM's mode
const char* agcn_chain_mode(void) { return agcn_chain_f16x3() ? "f16x3" : agcn_gemm_mode(); }
int agcn_conv_num_tiles(int V, int T_out) {
  int tt = agcn_conv_tile_frames(V, T_out);
  return (T_out + tt - 1) / tt;
}
// slots per sample of the (sum, sumsq) partials agcn_conv_fwd writes for these sizes (depends on the kernel picked)
int agcn_conv_stats_tiles(int Cin, int Cout, int T_out, int V, int taps, int stride) {
  (void)Cin; (void)stride;
  if (taps == 9 && agcn_chained() && Cout % 128 != 0 && agcn_bf16_conv_wide(taps, Cout)) {
    int tt = 512 / V;
    if (tt > T_out) tt = T_out;
    return (T_out + tt - 1) / tt;
  }
  return agcn_conv_num_tiles(V, T_out);
}
int agcn_dadj_num_slots(int C, int V, int T) {
  if (agcn_chained() && agcn_gcn_dadj_chain_supported(C, V)) return agcn_gcn_dadj_chain_slots(C, T);
  int tt = 128 / V;
  if (tt > T) tt = T;
  int ntiles = (T + tt - 1) / tt;
  int nmb = C >= 64 ? C / 64 : 1;
  return ntiles * nmb;
}

// bytes of workspace (packed weight images) the contraction entry points need; an upper bound over
// forward / backward-data of a conv with these sizes, and over the aggregate/dadj kernels with C=Cin
size_t agcn_conv_workspace(int Cin, int Cout, int T, int V, int taps, int stride) {
  const int pad = (taps - 1) / 2;
  const int To = (T + 2 * pad - taps) / stride + 1;
  size_t b = 0, t;
  if (taps == 9) {
    b = PACK_BYTES_BM(9, 0, CK9, CK9, V, To, stride, Cout, Cin);
    t = PACK_BYTES_BM(9, 0, CK9, CK9, V, T, 1, Cin, Cout); if (t > b) b = t;
    t = PACK_BYTES_BM(5, 0, CK9, CK9, V, (T + 1) / 2, 1, Cin, Cout); if (t > b) b = t;
    t = agcn_bf16_conv_workspace(Cin, Cout, T, V, stride); if (t > b) b = t;
  } else {
    b = PACK_BYTES_BM(1, 0, CK1, CK1, V, To, stride, Cout, Cin);
    t = PACK_BYTES_BM(1, 0, CK1, CK1, V, T, 1, Cin, Cout); if (t > b) b = t;
    t = agcn_bf16_conv1_workspace(Cin, Cout, T, V, stride); if (t > b) b = t;
  }
  return b + 256;
}
size_t agcn_gcn_workspace(int C, int Cout, int T, int V) {
  size_t b = PACK_BYTES_BM(1, 1, CKA, CKA, V, T, 1, Cout, C), t;
  t = PACK_BYTES_BM(1, 2, CKA, CKA, V, T, 1, C, Cout); if (t > b) b = t;
  t = pack_bytes<1, 0, 1, 4, 2, 1, CKD, 1>(V, T, 1, 3 * C, Cout); if (t > b) b = t;
  if (agcn_gcn_chain_supported(Cout, C, V)) { t = agcn_gcn_chain_workspace(Cout, C, 0, T, V); if (t > b) b = t; }
  // backward-data, with room for the fused theta/phi term (6*Cout/4 channels)
  if (agcn_gcn_chain_supported(C, Cout, V)) { t = agcn_gcn_chain_workspace(C, Cout, 6 * (Cout / 4), T, V); if (t > b) b = t; }
  if (agcn_gcn_dadj_chain_supported(C, V)) { t = agcn_gcn_dadj_chain_workspace(C, Cout); if (t > b) b = t; }
  return b + 256;
}

// slots per sample of the (sum, sumsq) partials agcn_gcn_aggregate_project_fwd writes for these sizes
int agcn_gcn_stats_tiles(int C, int Cout, int T, int V) {
  if (agcn_chained() && C >= 32 && agcn_gcn_chain_supported(Cout, C, V)) return agcn_gcn_chain_tiles(T);
  return agcn_conv_num_tiles(V, T);
}

// total slots (all samples) of the same partials: what the caller allocates (the persistent chain kernel writes one slot
// per (sample, frame split), whose count depends on N)
int agcn_gcn_stats_slots(int N, int C, int Cout, int T, int V) {
  if (agcn_chained() && C >= 32 && agcn_gcn_chain_supported(Cout, C, V)) return agcn_gcn_chain_stats_slots(N, Cout, C, T, V);
  return N * agcn_conv_num_tiles(V, T);
}

// y[n][o][t,v] = bias[o] + sum_{c,k} w[o][c][k] x[n][c][(t*stride + k - pad), v]      (unit_tcn conv, 1x1 convs)
int agcn_conv_fwd_ex(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* workspace,
                     size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                     const float* x_absmax, void* stream);
int agcn_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* workspace,
                  size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                  void* stream) {
  return agcn_conv_fwd_ex(x, w, bias, y, stats_part, workspace, workspace_bytes, N, Cin, Cout, T, V, taps, stride, nullptr,
                          stream);
}

// same; x_absmax (optional): device scalar max |x| left by agcn_bn_act_fwd_ex when it produced x -- saves the split-fp16
// temporal convolution its own pass over x for the range scale
int agcn_conv_fwd_ex(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* workspace,
                     size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                     const float* x_absmax, void* stream) {
  if (!x || !w || !y || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if ((taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return AGCN_ERR_UNSUPPORTED;
  const int pad = (taps - 1) / 2;
  if (taps == 9 && agcn_gemm_precision() != 0)
    return agcn_bf16_conv9_fwd(x, w, bias, y, stats_part, workspace, workspace_bytes, N, Cin, Cout, T, V, stride,
                               agcn_gemm_precision(), (hipStream_t)stream, nullptr, 0, x_absmax);
  Problem p = {};
  ConvGemmArgs& a = p.a;
  a.in = x; a.bias = bias; a.out = y; a.stats = stats_part;
  a.N = N; a.M = Cout; a.Kinner = Cin; a.in_rows = Cin; a.V = V;
  a.T_src = T; a.T_out = (T + 2 * pad - taps) / stride + 1; a.T_full = a.T_out;
  a.src_stride = stride; a.f_off = -pad; a.out_fs = 1; a.out_fo = 0;
  p.w = w; p.sa_m = (long)Cin * taps; p.sa_i = 0; p.sa_c = taps; p.tap_flip_from = -1;
  p.ws = workspace; p.ws_bytes = workspace_bytes;
  hipStream_t s = (hipStream_t)stream;
  if (taps == 9) return DISPATCH_BM(9, 0, CK9, CK9, 11, p, s);
  if (stride == 1) return DISPATCH_BM(1, 0, CK1, CK1, 4, p, s);
  return DISPATCH_BM(1, 0, CK1, CK1, 8, p, s);
}

// dx[n][c][t,v] (+)= sum_{o,k} w[o][c][k] dy[n][o][(t + pad - k)/stride, v]  (+ masked addends)
int agcn_conv_bwd_data_ex(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                          const float* mask1, const float* add2, const float* mask2, void* workspace,
                          size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                          const float* dy_absmax, void* stream);
int agcn_conv_bwd_data(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                       const float* mask1, const float* add2, const float* mask2, void* workspace,
                       size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                       void* stream) {
  return agcn_conv_bwd_data_ex(dy, w, dx, accumulate, add1, mask1, add2, mask2, workspace, workspace_bytes, N, Cin, Cout, T,
                               V, taps, stride, nullptr, stream);
}

// same; dy_absmax (optional): device scalar max |dy| left by agcn_bn_bwd_apply_ex
int agcn_conv_bwd_data_ex(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                          const float* mask1, const float* add2, const float* mask2, void* workspace,
                          size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                          const float* dy_absmax, void* stream) {
  if (!dy || !w || !dx || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if ((taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return AGCN_ERR_UNSUPPORTED;
  const int pad = (taps - 1) / 2;
  if (taps == 9 && agcn_gemm_precision() != 0)
    return agcn_bf16_conv9_bwd_data(dy, w, dx, accumulate, add1, mask1, add2, mask2, workspace, workspace_bytes, N, Cin,
                                    Cout, T, V, stride, agcn_gemm_precision(), (hipStream_t)stream, dy_absmax);
  // 1x1 backward-data: measured 10-17% faster on the split-bf16 kernel; the 1x1 forward (store-bound) is not
  if (taps == 1 && stride == 1 && agcn_chained() && Cin >= 64 && Cout >= 32)
    return agcn_bf16_conv1_bwd_data(dy, w, dx, accumulate, add1, mask1, add2, mask2, workspace, workspace_bytes, N, Cin,
                                    Cout, T, V, agcn_gemm_precision(), (hipStream_t)stream);
  Problem p = {};
  ConvGemmArgs& a = p.a;
  a.in = dy; a.out = dx; a.accumulate = accumulate;
  a.add1 = add1; a.mask1 = mask1; a.add2 = add2; a.mask2 = mask2;
  a.N = N; a.M = Cin; a.Kinner = Cout; a.in_rows = Cout; a.V = V;
  a.T_src = (T + 2 * pad - taps) / stride + 1; a.T_full = T; a.src_stride = 1;
  p.w = w; p.sa_m = taps; p.sa_i = 0; p.sa_c = (long)Cin * taps;
  p.ws = workspace; p.ws_bytes = workspace_bytes;
  hipStream_t s = (hipStream_t)stream;
  if (stride == 1) {
    // dx[t] = sum_j W[taps-1-j] dy[t + j - pad]
    a.T_out = T; a.out_fs = 1; a.out_fo = 0; a.f_off = -pad;
    p.tap_mul = 1; p.tap_add = 0; p.tap_flip_from = taps - 1;
    if (taps == 9) return DISPATCH_BM(9, 0, CK9, CK9, 8, p, s);
    return DISPATCH_BM(1, 0, CK1, CK1, 4, p, s);
  }
  // stride 2: output frames of parity `par` form a stride-1 problem over tau (t = 2*tau + par):
  //   dx[2tau+par] = sum_j W[k = taps-1-(2j+par')] dy[tau + j + off]   with only the taps of matching parity
  int rc;
  if (taps == 9) {
    // even t: tap' = 2j (j<5), source tau + j - 2 ; odd t: tap' = 2j+1 (j<4), source tau + j - 1
    a.T_out = (T + 1) / 2; a.out_fs = 2; a.out_fo = 0; a.f_off = -2;
    p.tap_mul = 2; p.tap_add = 0; p.tap_flip_from = 8;
    rc = DISPATCH_BM(5, 0, CK9, CK9, 8, p, s);
    if (rc) return rc;
    a.T_out = T / 2; a.out_fo = 1; a.f_off = -1;
    p.tap_add = 1;
    if (a.T_out > 0) rc = DISPATCH_BM(4, 0, CK9, CK9, 8, p, s);
    return rc;
  }
  // 1x1 stride 2: even t takes dy[t/2]; odd t receives no signal (zero + addends)
  a.T_out = (T + 1) / 2; a.out_fs = 2; a.out_fo = 0; a.f_off = 0;
  p.tap_mul = 1; p.tap_add = 0; p.tap_flip_from = 0;
  rc = DISPATCH_BM(1, 0, CK1, CK1, 4, p, s);
  if (rc) return rc;
  a.T_out = T / 2; a.out_fo = 1; a.Kinner = 0;      // no K chunks: the epilogue writes 0 (+accumulate/addends)
  if (a.T_out > 0) rc = DISPATCH_BM(1, 0, CK1, CK1, 4, p, s);
  return rc;
}

// y[n][o][t,v] = bias[o] + sum_i sum_c wcat[o][i*C+c] * sum_u x[n][c][t,u] adj[n][i][u][v]
int agcn_gcn_aggregate_project_fwd_ex(const float* x, const float* adj, const float* wcat, const float* bias, float* y,
                                      float* stats_part, void* workspace, size_t workspace_bytes, int N, int C, int Cout,
                                      int T, int V, const float* x_absmax, void* stream);
int agcn_gcn_aggregate_project_fwd(const float* x, const float* adj, const float* wcat, const float* bias, float* y,
                                   float* stats_part, void* workspace, size_t workspace_bytes, int N, int C, int Cout,
                                   int T, int V, void* stream) {
  return agcn_gcn_aggregate_project_fwd_ex(x, adj, wcat, bias, y, stats_part, workspace, workspace_bytes, N, C, Cout, T, V,
                                           nullptr, stream);
}
// x_absmax: device scalar max |x| left behind by the kernel that produced x (agcn_bn_act_fwd_ex), for the range scale
// of the f16x3 chain; null: the chain takes it with a streaming pass of its own
int agcn_gcn_aggregate_project_fwd_ex(const float* x, const float* adj, const float* wcat, const float* bias, float* y,
                                      float* stats_part, void* workspace, size_t workspace_bytes, int N, int C, int Cout,
                                      int T, int V, const float* x_absmax, void* stream) {
  if (!x || !adj || !wcat || !y || !workspace || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  // (the 3-channel first layer's forward stays on the f32 kernel: measured 193 us against 260 us chained)
  if (agcn_chained() && C >= 32 && agcn_gcn_chain_supported(Cout, C, V))
    return agcn_gcn_chain(0, x, adj, wcat, bias, y, stats_part, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                          0, workspace, workspace_bytes, N, C, Cout, T, V, (hipStream_t)stream, 0, 0, x_absmax, nullptr);
  Problem p = {};
  ConvGemmArgs& a = p.a;
  a.in = x; a.bias = bias; a.out = y; a.adj = adj; a.stats = stats_part;
  a.N = N; a.M = Cout; a.Kinner = C; a.in_rows = C; a.V = V; a.T_src = T; a.T_out = T; a.T_full = T;
  a.src_stride = 1; a.f_off = 0; a.out_fs = 1; a.out_fo = 0;
  p.w = wcat; p.sa_m = 3L * C; p.sa_i = C; p.sa_c = 1; p.tap_flip_from = -1;
  p.ws = workspace; p.ws_bytes = workspace_bytes;
  return DISPATCH_BM(1, 1, CKA, CKA, 4, p, (hipStream_t)stream);
}

// ---- BN-folded inference (eval mode): the BatchNorm that follows a contraction is folded into its weights and bias
// by the caller; the residual add and the ReLU ride in the store epilogue, so a unit is adjacency + two kernels ----
// y = act( bias + sum_i W_i (x . adj_i) [+ res] [+ W2 . x2] ),  W2 (Cout, K2) row-major (a folded 1x1 `down` conv on x2)
// Only on the chained split-bf16 path (C >= 32): AGCN_ERR_UNSUPPORTED otherwise (the caller runs the unfused passes).
int agcn_gcn_unit_infer(const float* x, const float* adj, const float* wcat, const float* bias, const float* res,
                        const float* x2, const float* w2, int K2, int relu, float* y, void* workspace,
                        size_t workspace_bytes, int N, int C, int Cout, int T, int V, void* stream) {
  if (!x || !adj || !wcat || !y || !workspace || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if ((x2 == nullptr) != (w2 == nullptr) || (x2 && K2 <= 0)) return AGCN_ERR_ARG;
  if (!(agcn_chained() && C >= 32 && agcn_gcn_chain_supported(Cout, C, V))) return AGCN_ERR_UNSUPPORTED;
  if (x2 && K2 % 32 != 0) return AGCN_ERR_UNSUPPORTED;
  if (agcn_gcn_chain_workspace(Cout, C, x2 ? K2 : 0, T, V) > workspace_bytes) return AGCN_ERR_WORKSPACE;
  return agcn_gcn_chain(0, x, adj, wcat, bias, y, nullptr, 0, res, nullptr, nullptr, nullptr, 0, x2, w2, x2 ? K2 : 0,
                        workspace, workspace_bytes, N, C, Cout, T, V, (hipStream_t)stream, relu, 1);
}

size_t agcn_gcn_unit_infer_workspace(int C, int Cout, int K2, int T, int V) {
  return agcn_gcn_chain_workspace(Cout, C, K2, T, V) + 256;
}

// y = act( bias + conv9x1(x; w, stride) [+ res] ),  res (N, Cout, T_out, V).  Split-bf16 modes only.
int agcn_conv9_infer(const float* x, const float* w, const float* bias, const float* res, int relu, float* y,
                     void* workspace, size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int stride,
                     void* stream) {
  if (!x || !w || !y || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if (stride != 1 && stride != 2) return AGCN_ERR_UNSUPPORTED;
  if (agcn_gemm_precision() == 0) return AGCN_ERR_UNSUPPORTED;
  return agcn_bf16_conv9_fwd(x, w, bias, y, nullptr, workspace, workspace_bytes, N, Cin, Cout, T, V, stride,
                             agcn_gemm_precision(), (hipStream_t)stream, res, relu);
}

// dx[n][c][t,u] (+)= sum_i sum_o wcat[o][i*C+c] * sum_v dy[n][o][t,v] adj[n][i][u][v]   (+ masked addends)
int agcn_gcn_bwd_data_fused_supported(int C, int Cout, int V);
int agcn_gcn_aggregate_project_bwd_data_ex(const float* dy, const float* adj, const float* wcat, const float* dtp,
                                           const float* w2, int K2, float* dx, int accumulate, const float* add1,
                                           const float* mask1, const float* add2, const float* mask2, int mask_bits,
                                           void* workspace, size_t workspace_bytes, int N, int C, int Cout, int T, int V,
                                           const float* dy_absmax, const float* dtp_absmax, void* stream);
int agcn_gcn_aggregate_project_bwd_data(const float* dy, const float* adj, const float* wcat, float* dx,
                                        int accumulate, const float* add1, const float* mask1, const float* add2,
                                        const float* mask2, int mask_bits, void* workspace, size_t workspace_bytes,
                                        int N, int C, int Cout, int T, int V, void* stream) {
  return agcn_gcn_aggregate_project_bwd_data_ex(dy, adj, wcat, nullptr, nullptr, 0, dx, accumulate, add1, mask1, add2,
                                                mask2, mask_bits, workspace, workspace_bytes, N, C, Cout, T, V, nullptr,
                                                nullptr, stream);
}
// Both backward-data forms behind one entry point (dtp null: the plain one; else the fused 1x1 term as in
// agcn_gcn_aggregate_project_bwd_data_fused), with the device scalars max |dy| / max |dtp| their producers left behind
// (agcn_bn_bwd_apply_ex; null: the f16x3 chain takes them with a streaming pass of its own).
int agcn_gcn_aggregate_project_bwd_data_ex(const float* dy, const float* adj, const float* wcat, const float* dtp,
                                           const float* w2, int K2, float* dx, int accumulate, const float* add1,
                                           const float* mask1, const float* add2, const float* mask2, int mask_bits,
                                           void* workspace, size_t workspace_bytes, int N, int C, int Cout, int T, int V,
                                           const float* dy_absmax, const float* dtp_absmax, void* stream) {
  if (!dy || !adj || !wcat || !dx || !workspace || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if (dtp) {
    if (!w2 || K2 <= 0) return AGCN_ERR_ARG;
    if (!agcn_gcn_bwd_data_fused_supported(C, Cout, V)) return AGCN_ERR_UNSUPPORTED;
    if (agcn_gcn_chain_workspace(C, Cout, K2, T, V) > workspace_bytes) return AGCN_ERR_WORKSPACE;
    return agcn_gcn_chain(1, dy, adj, wcat, nullptr, dx, nullptr, accumulate, add1, mask1, add2, mask2, mask_bits, dtp,
                          w2, K2, workspace, workspace_bytes, N, C, Cout, T, V, (hipStream_t)stream, 0, 0, dy_absmax,
                          dtp_absmax);
  }
  if (agcn_chained() && agcn_gcn_chain_supported(C, Cout, V))
    return agcn_gcn_chain(1, dy, adj, wcat, nullptr, dx, nullptr, accumulate, add1, mask1, add2, mask2, mask_bits,
                          nullptr, nullptr, 0, workspace, workspace_bytes, N, C, Cout, T, V, (hipStream_t)stream, 0, 0,
                          dy_absmax, nullptr);
  Problem p = {};
  ConvGemmArgs& a = p.a;
  a.in = dy; a.out = dx; a.adj = adj; a.accumulate = accumulate;
  a.add1 = add1; a.mask1 = mask1; a.add2 = add2; a.mask2 = mask2; a.mask_bits = mask_bits;
  a.N = N; a.M = C; a.Kinner = Cout; a.in_rows = Cout; a.V = V; a.T_src = T; a.T_out = T; a.T_full = T;
  a.src_stride = 1; a.f_off = 0; a.out_fs = 1; a.out_fo = 0;
  p.w = wcat; p.sa_m = 1; p.sa_i = C; p.sa_c = 3L * C; p.tap_flip_from = -1;
  p.ws = workspace; p.ws_bytes = workspace_bytes;
  return DISPATCH_BM(1, 2, CKA, CKA, 4, p, (hipStream_t)stream);
}

// Same, plus the 1x1 term of the adaptive branch in one pass:  dx (+)= ... + W2^T dtp  with dtp (N, K2, T, V) and
// w2 (K2, C) row-major (the stacked conv_a/conv_b weights).  Only on the chained (bf16x6) path: check
// agcn_gcn_bwd_data_fused_supported first.
int agcn_gcn_bwd_data_fused_supported(int C, int Cout, int V) {
  return agcn_chained() && agcn_gcn_chain_supported(C, Cout, V) ? 1 : 0;
}
int agcn_gcn_aggregate_project_bwd_data_fused(const float* dy, const float* adj, const float* wcat, const float* dtp,
                                              const float* w2, int K2, float* dx, int accumulate, const float* add1,
                                              const float* mask1, const float* add2, const float* mask2, int mask_bits,
                                              void* workspace, size_t workspace_bytes, int N, int C, int Cout, int T,
                                              int V, void* stream) {
  if (!dtp) return AGCN_ERR_ARG;
  return agcn_gcn_aggregate_project_bwd_data_ex(dy, adj, wcat, dtp, w2, K2, dx, accumulate, add1, mask1, add2, mask2,
                                                mask_bits, workspace, workspace_bytes, N, C, Cout, T, V, nullptr, nullptr,
                                                stream);
}

// dadj_part[n][i][slot][u][v] = sum over the slot's (c,t) of x[n][c][t,u] * (sum_o wcat[o][i*C+c] dy[n][o][t,v])
int agcn_gcn_dadj_ex(const float* dy, const float* wcat, const float* x, float* dadj_part, void* workspace,
                     size_t workspace_bytes, int N, int C, int Cout, int T, int V, const float* dy_absmax,
                     const float* x_absmax, void* stream);
int agcn_gcn_dadj(const float* dy, const float* wcat, const float* x, float* dadj_part, void* workspace,
                  size_t workspace_bytes, int N, int C, int Cout, int T, int V, void* stream) {
  return agcn_gcn_dadj_ex(dy, wcat, x, dadj_part, workspace, workspace_bytes, N, C, Cout, T, V, nullptr, nullptr, stream);
}
// dy_absmax: device scalar max |dy| (agcn_bn_bwd_apply_ex) for the f16x3 projection; null: a streaming pass inside.
// x_absmax: device scalar max |x| (kept by the forward) for the f16x3 reduction against x; null: a streaming pass inside
int agcn_gcn_dadj_ex(const float* dy, const float* wcat, const float* x, float* dadj_part, void* workspace,
                     size_t workspace_bytes, int N, int C, int Cout, int T, int V, const float* dy_absmax,
                     const float* x_absmax, void* stream) {
  if (!dy || !wcat || !x || !dadj_part || !workspace || N <= 0 || C <= 0 || Cout <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if (agcn_chained() && agcn_gcn_dadj_chain_supported(C, V))
    return agcn_gcn_dadj_chain(dy, wcat, x, dadj_part, workspace, workspace_bytes, N, C, Cout, T, V, (hipStream_t)stream,
                               dy_absmax, x_absmax);
  if (C >= 64 && C % 64 != 0) return AGCN_ERR_UNSUPPORTED;
  Problem p = {};
  ConvGemmArgs& a = p.a;
  a.in = dy; a.xin = x; a.dadj = dadj_part;
  a.N = N; a.M = 3 * C; a.Kinner = Cout; a.in_rows = Cout; a.V = V; a.T_src = T; a.T_out = T; a.T_full = T;
  a.src_stride = 1; a.f_off = 0; a.out_fs = 1; a.out_fo = 0; a.C = C;
  p.w = wcat; p.sa_m = 1; p.sa_i = 0; p.sa_c = 3L * C; p.tap_flip_from = -1;
  p.ws = workspace; p.ws_bytes = workspace_bytes;
  return launch_cfg<1, 0, 1, 4, 2, 1, CKD, 2, 1>(p, (hipStream_t)stream);
}

}  // extern "C"
