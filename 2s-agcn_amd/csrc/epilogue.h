// Coalesced store epilogue shared by the contraction kernels.
// The MFMA accumulator layout gives every lane 16 rows x 1 column, so a direct store touches 32-float (128 B) runs of
// many rows at once.  Instead the workgroup's (BM x positions) tile is passed through LDS (the caller fills `tile`
// and synchronises) and written row-wise: a row is one contiguous run (or V-float runs for frame-strided outputs),
// its residual / accumulate operands are fetched with the same coalescing, and the per-channel (sum, sumsq) partials
// of the BatchNorm that follows are reduced over LDS columns (fixed order: bitwise reproducible).
#pragma once
#include "agcn_common.h"

struct EpiPtrs {
  float* out;
  const float* add1;
  const float* mask1;
  const float* add2;
  const float* mask2;
  float* stats;          // [slot][2][M] or null
  int accumulate;
  int relu;              // out = max(., 0) (BN-folded inference)
};

// tile  : [BM][TP] (TP odd) valid for columns q < nvalid; bias_s: [BM] in LDS; red: [NT*2] floats of LDS scratch
// row r of the tile is output row m0 + r; element q of a row lives at out[row_base(m) + poff(q)] where
// row_base(m) = rows0 + m*P and poff is given per lane for q = lane + 64*u.
template <int BM, int NW, int XB>
__device__ __forceinline__ void epilogue_rows(const EpiPtrs& e, const float* tile, int TP, const float* bias_s,
                                              float* red, int M, int m0, long rows0, long P, int nvalid,
                                              const int (&poff)[XB], long stats_slot) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Row groups are software-pipelined: the operands of group k+1 are loaded RAW into registers while group k is
  // combined and stored.
  constexpr int RG = (XB >= 8) ? 1 : ((XB >= 4) ? 2 : 4), NG = BM / (NW * RG);   // rows in flight per wave (register budget)
  static_assert(BM % (NW * RG) == 0, "row groups must tile the block");
  const bool has_extra = e.accumulate || e.add1 || e.add2;   // kernel-uniform
  float ex[2][5][RG][XB];                                    // [buffer][out, add1, mask1, add2, mask2]
  auto row_base = [&](int k, int g) __attribute__((always_inline)) {
    const int m = min(m0 + (k * NW + wave) * RG + g, M - 1);
    return rows0 + (long)m * P;
  };
  auto load_extras = [&](int k, float (&x)[5][RG][XB]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const long base = row_base(k, g);
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        const long idx = base + ((q < nvalid) ? poff[u] : 0);
        if (e.accumulate) x[0][g][u] = e.out[idx];
        if (e.add1) x[1][g][u] = e.add1[idx];
        if (e.mask1) x[2][g][u] = e.mask1[idx];
        if (e.add2) x[3][g][u] = e.add2[idx];
        if (e.mask2) x[4][g][u] = e.mask2[idx];
      }
    }
  };
  if (has_extra) load_extras(0, ex[0]);
  if (e.stats) {
    // per-channel (sum, sumsq) of y = tile + bias over the valid positions: thread <-> (row, column slice)
    constexpr int NP = NT / BM;
    static_assert(NT % BM == 0, "threads must tile the rows");
    const int r = tid % BM, part = tid / BM;
    const float bval = bias_s[r];
    float bsum = 0.f, bsq = 0.f;
#pragma unroll 8
    for (int q = part; q < nvalid; q += NP) {
      const float y = tile[r * TP + q] + bval;
      bsum += y;
      bsq += y * y;
    }
    red[(part * 2 + 0) * BM + r] = bsum;
    red[(part * 2 + 1) * BM + r] = bsq;
    __syncthreads();
    for (int i = tid; i < 2 * BM; i += NT) {
      const int k = i / BM, ml = i - k * BM;
      float sum = 0.f;
#pragma unroll
      for (int p2 = 0; p2 < NP; ++p2) sum += red[(p2 * 2 + k) * BM + ml];
      if (m0 + ml < M) e.stats[(stats_slot * 2 + k) * M + m0 + ml] = sum;
    }
  }
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int r0 = (k * NW + wave) * RG;
    if (m0 + r0 >= M) break;                           // wave-uniform
    if (has_extra && k + 1 < NG) load_extras(k + 1, ex[(k + 1) & 1]);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const int m = m0 + r0 + g;
      if (m >= M) break;                               // wave-uniform
      const long base = row_base(k, g);
      const float bval = bias_s[r0 + g];
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        float v = tile[(r0 + g) * TP + min(q, TP - 1)] + bval;
        if (e.accumulate) v += ex[k & 1][0][g][u];
        if (e.add1) v += (!e.mask1 || ex[k & 1][2][g][u] > 0.f) ? ex[k & 1][1][g][u] : 0.f;
        if (e.add2) v += (!e.mask2 || ex[k & 1][4][g][u] > 0.f) ? ex[k & 1][3][g][u] : 0.f;
        if (e.relu) v = fmaxf(v, 0.f);
        if (q < nvalid) e.out[base + poff[u]] = v;
      }
    }
  }
}
