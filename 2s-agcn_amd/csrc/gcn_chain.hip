// unit_gcn's fused aggregate+project  y = sum_i Wd_i (x . A^_i)  (reference agcn.py:103-105) and its backward-data
// dx = sum_i Wd_i^T (dy . A^_i^T), as a register-chained pair of matrix-core contractions:
//
//   wave  <->  one frame t (V <= 32 joints = one 32-column MFMA tile), all BM output rows of the block
//   stage <->  (32-channel block cb, subset i)
//   1. G = X[cb](32 channels x V) . A^_i (V x V)      exact-f32 MFMA chain (v_mfma_f32_32x32x2_f32), operands from LDS
//      (x chunk staged once per cb, the sample's three padded adjacencies staged once per workgroup).
//   2. The D registers of step 1 ARE the B operand of the projection: register j of lane (h, v) holds channel
//      (j&3) + 8*(j>>2) + 4*h, so the 16 values of a lane are split in registers into bf16 (hi, mid, lo) pieces and
//      fed to v_mfma_f32_32x32x16_bf16 (bf16x6: 6 products per fp32 product, fp32 accumulate = fp32-equivalent, see
//      conv_gemm_bf16.hip) against weight images the pack kernel pre-split and pre-permuted to the same channel order.
//   G never touches LDS or HBM, there is no aggregate barrier, and no wave idles on an uneven tile count.
//
// Pipeline: weight images are double-buffered in LDS; the global loads of stage s+2 (and of the next x chunk) are
// issued into registers before the matrix-core work of stage s and committed before stage s+1: one barrier per stage.
#include "agcn_common.h"
#include "split_f16.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));     // activation rows are only 4-byte aligned

constexpr int CB = 32;                    // channels per stage

struct ChainArgs {
  const float* in;             // (N, K, T, V): x (forward) or dy (backward-data)
  const unsigned short* wp;    // packed weight images [mblock][cb][i][plane][hf][tm][lane][8] (bf16)
  const float* bias;
  float* out;                  // (N, M, T, V)
  const float* adj;            // (N, 3, V, V)
  float* stats;                // [N*ntiles][2][M] or null
  const float* add1;
  const float* mask1;
  const float* add2;
  const float* mask2;
  int N, M, K, T, V;
  int ntiles, ncb, nmb;
  int accumulate;
  int adj_t;                   // 0: B[u][v] = adj[u][v] (forward); 1: B[u][v] = adj[v][u] (backward-data)
  int mask_bits;               // mask1/mask2 are sign bit masks (agcn_bn_act_fwd) instead of fp32 tensors
  const float* in2;            // optional second, un-aggregated source (N, K2, T, V): acc += W2 . in2 (plain stages)
  int K2, ncb2;                // its channels and 32-channel blocks (0: none)
  int XP;                      // pitch (floats) of an x chunk row in LDS (odd)
  int off_bias;                // byte offset of the bias row in LDS
  int npl;                     // 3: six split products (bf16x6) ; 1: hi*hi only (bf16)
  int dbg;                     // profiling switches (AGCN_GC_DBG): 1 = no matrix work, 2 = no staging after the prologue
  int relu;                    // epilogue: out = max(., 0) (BN-folded inference)
  const float* in_absmax;      // f16x3: device scalars max |in|, max |in2| (null: none) for the range scale
  const float* in2_absmax;
};

struct ChainPackArgs {
  const float* w;
  unsigned short* wp;
  int M, K, ncb;
  long sa_m, sa_i, sa_c;       // W_i[m][c] = w[m*sa_m + i*sa_i + c*sa_c]
  int nsub, s_total, s_off;    // subsets per channel block (3 / 1), stage images per row block, first stage written
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 p = __builtin_convertvector(v, bf16x2);      // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float lo_as_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_as_f32(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ void split_pair(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
  ph = pack_bf16(a, b);
  const float ra = a - lo_as_f32(ph), rb = b - hi_as_f32(ph);
  pm = pack_bf16(ra, rb);
  pl = pack_bf16(ra - lo_as_f32(pm), rb - hi_as_f32(pm));
}

// channel (within the 32-block) that slot e of lane-half h carries in projection half hf: D register j = 8*hf + e
__host__ __device__ __forceinline__ int chain_channel(int hf, int h, int e) {
  const int j = 8 * hf + e;
  return (j & 3) + 8 * (j >> 2) + 4 * h;
}

// one block per (mblock, cb, i) image: [plane][hf][tm][lane][8]
// F16: two fp16 planes (f16x3) instead of three bf16 planes
template <int TM, bool F16>
__global__ void __launch_bounds__(256) chain_pack_kernel(const ChainPackArgs p) {
  constexpr int BM = TM * 32;
  constexpr int PL = F16 ? 2 : 3;
  constexpr int PER_PLANE = 2 * TM * 64 * 8;          // 16-bit elements
  const int i = blockIdx.x % p.nsub;
  const int cb = (blockIdx.x / p.nsub) % p.ncb;
  const int mb = blockIdx.x / (p.nsub * p.ncb);
  unsigned short* dst = p.wp + ((long)mb * p.s_total + p.s_off + (long)cb * p.nsub + i) * PL * PER_PLANE;
  for (int e = threadIdx.x; e < PER_PLANE / 2; e += 256) {     // one bf16 pair per iteration
    const int e2 = e & 3;                 // slot pair (2*e2, 2*e2+1)
    const int lane = (e >> 2) & 63;
    const int r = e >> 8;
    const int tm = r % TM, hf = r / TM;
    const int h = lane >> 5, lr = lane & 31;
    const int m = mb * BM + tm * 32 + lr;
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = cb * CB + chain_channel(hf, h, 2 * e2 + q);
      v[q] = (m < p.M && c < p.K) ? p.w[(long)m * p.sa_m + (long)i * p.sa_i + (long)c * p.sa_c] : 0.f;
    }
    unsigned ph, pm, pl = 0;
    if constexpr (F16) split_pair_f16(v[0] * F16_W_SCALE, v[1] * F16_W_SCALE, ph, pm);
    else split_pair(v[0], v[1], ph, pm, pl);
    const int o = ((hf * TM + tm) * 64 + lane) * 8 + 2 * e2;
    *reinterpret_cast<unsigned*>(dst + 0 * PER_PLANE + o) = ph;
    *reinterpret_cast<unsigned*>(dst + 1 * PER_PLANE + o) = pm;
    if constexpr (!F16) *reinterpret_cast<unsigned*>(dst + 2 * PER_PLANE + o) = pl;
  }
}

// VS >= (V+1)/2 aggregation steps (13: NTU V=25, 9: Kinetics V=18, 16: any V <= 32; surplus steps multiply zero rows)
// NW waves per workgroup = frames per workgroup tile (one frame per wave)
// F16 (with VS == 0): both contractions on f16x3 (split_f16.h) instead of bf16x6: two fp16 planes per operand, three
// products; the staged source is multiplied by the power of two that brings max(|in|, |in2|) into [2^2, 2^3)
// (F16_ADJ_TARGET) before it is split (so that G = x . A^ stays inside fp16's range for column sums of |A^| up to 2^13)
// and the accumulators by its inverse in the epilogue.
template <int TM, int VS, int NW, bool F16 = false>
__global__ void __launch_bounds__(NW * 64, 2) gcn_chain_kernel(const ChainArgs a) {
  static_assert(!F16 || VS == 0, "the f16x3 chain runs the aggregation on split MFMA too");
  constexpr int NT = NW * 64, FT = NW;
  constexpr int BM = TM * 32;
  constexpr int PL = F16 ? 2 : 3;                      // planes per split operand
  constexpr int NPROD = F16 ? 3 : 6;                   // MFMA products per step
  constexpr int A_IMG = PL * 2 * TM * 1024;            // bytes of one stage's weight image
  constexpr int A16 = A_IMG / 16;
  constexpr int EA = (A16 + NT - 1) / NT;
  constexpr int XB = FT / 2;                           // 64-float column blocks of a staged x row (V <= 32)
  constexpr int XR = CB / NW;                          // x rows per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // VS == 0: the aggregation runs on split-bf16 MFMA too (K = 32 joints = two 16-deep steps, 12 MFMAs of 32 cycles
  // instead of 13 exact-f32 steps of 64): the adjacencies are kept as bf16 planes [i][plane][ks][h][v][8 u], a lane's
  // x fragment (8 consecutive joints of its channel) is split in registers once per channel block and serves all
  // three subsets.
  constexpr bool BCH = (VS == 0);
  constexpr int ADJ_BYTES = BCH ? 3 * PL * 2 * 2 * 32 * 16 : 3 * 32 * 32 * 4;
  float* adjp = reinterpret_cast<float*>(smem);                       // [3][32][32] (f32 chain)
  unsigned char* adjq = smem;                                         // bf16 planes (split chain)
  float* xb = reinterpret_cast<float*>(smem + ADJ_BYTES);             // [CB][XP] (+ 32 floats of slack when BCH)
  const int xb_bytes = ((CB * a.XP * 4 + (BCH ? 128 : 0) + 15) & ~15);
  unsigned char* abuf = smem + ADJ_BYTES + xb_bytes;                  // [2][A_IMG]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mbk = bid % a.nmb;
  const int nt_id = bid / a.nmb;
  const int n = nt_id / a.ntiles, tile_id = nt_id - n * a.ntiles;
  const int m0 = mbk * BM;
  const int V = a.V, T = a.T, XP = a.XP;
  const int t0 = tile_id * FT;
  const int t = t0 + wave;
  const bool fvalid = t < T;                           // wave-uniform
  const int xlen = min(FT, T - t0) * V;                // valid floats of a staged x row
  const long P = (long)T * V;
  const int S1 = 3 * a.ncb;                            // aggregated stages, then a.ncb2 plain stages
  const int S = S1 + a.ncb2;
  const int nchunks = a.ncb + a.ncb2;                  // staged source chunks: x blocks, then in2 blocks
  float rs_s = 1.f, rs_inv = 1.f;                      // f16x3 range scale of the staged sources
  if constexpr (F16) {
    float mx = a.in_absmax ? *a.in_absmax : 0.f;
    if (a.in2_absmax) mx = fmaxf(mx, *a.in2_absmax);
    f16_range_scale_of<F16_ADJ_TARGET>(mx, rs_s, rs_inv);
    rs_inv *= F16_W_INV;                               // (the packed weights carry F16_W_SCALE)
  }

  // bias of this row block, beyond everything the epilogue tile overwrites
  float* bias_s = reinterpret_cast<float*>(smem + a.off_bias);
  for (int e = tid; e < BM; e += NT) bias_s[e] = (a.bias && m0 + e < a.M) ? a.bias[m0 + e] : 0.f;

  // ---- the sample's adjacencies, zero padded to 32x32 (rows u >= V are zero: clamped x operands need no mask) ----
  if (!BCH) {
    const float* adjn = a.adj + (long)n * 3 * V * V;
    for (int e = tid; e < 3 * 32 * 32; e += NT) {
      const int i = e >> 10, u = (e >> 5) & 31, col = e & 31;
      const bool ok = u < V && col < V;
      const int gi = ok ? (a.adj_t ? ((i * V + col) * V + u) : ((i * V + u) * V + col)) : 0;
      const float tv = adjn[gi];
      adjp[e] = ok ? tv : 0.f;
    }
  } else {
    // B[u][v] of subset i as the B operand of v_mfma_f32_32x32x16_bf16: lane (h, v) holds u = 16 ks + 8 h + e
    const float* adjn = a.adj + (long)n * 3 * V * V;
    for (int e = tid; e < 3 * 2 * 2 * 32 * 4; e += NT) {             // one pair (u, u+1) per iteration
      const int e2 = e & 3, col = (e >> 2) & 31, hh = (e >> 7) & 1, ks = (e >> 8) & 1, i = e >> 9;
      float v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int u = 16 * ks + 8 * hh + 2 * e2 + q;
        const bool ok = u < V && col < V;
        const int gi = ok ? (a.adj_t ? ((i * V + col) * V + u) : ((i * V + u) * V + col)) : 0;
        const float tv = adjn[gi];
        v[q] = ok ? tv : 0.f;
      }
      unsigned p0, p1, p2 = 0;
      if constexpr (F16) split_pair_f16(v[0], v[1], p0, p1);
      else split_pair(v[0], v[1], p0, p1, p2);
      const int o = (((ks * 2 + hh) * 32) + col) * 16 + e2 * 4;     // within one (i, plane) image of 2048 bytes
      *reinterpret_cast<unsigned*>(adjq + (i * PL + 0) * 2048 + o) = p0;
      *reinterpret_cast<unsigned*>(adjq + (i * PL + 1) * 2048 + o) = p1;
      if constexpr (!F16) *reinterpret_cast<unsigned*>(adjq + (i * PL + 2) * 2048 + o) = p2;
    }
    // slack behind the last chunk row: the last frame's 32-joint fragment of the last row runs up to 32 - V floats past it
    // (they meet zero rows of the adjacency, but must be finite: stale LDS bits can be NaN patterns)
    if (tid < 32) xb[CB * XP + tid] = 0.f;
  }

  f32x16 acc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;
  if (XP > FT * V && tid < CB) xb[tid * XP + FT * V] = 0.f;   // the pad element of a chunk row is read (times zero)

  // ---- prefetch registers (raw loads; predicates are re-evaluated at commit) ----
  u32x4 ra[EA];
  float rx[XR][XB];
  const u32x4* wp4 = reinterpret_cast<const u32x4*>(a.wp) + (long)mbk * S * A16;
  auto issue_A = [&](int s) __attribute__((always_inline)) {
    const u32x4* src = wp4 + (long)s * A16;
#pragma unroll
    for (int u = 0; u < EA; ++u) ra[u] = src[min(tid + u * NT, A16 - 1)];
  };
  auto commit_A = [&](int s) __attribute__((always_inline)) {
    u32x4* dst = reinterpret_cast<u32x4*>(abuf + (s & 1) * A_IMG);
#pragma unroll
    for (int u = 0; u < EA; ++u)
      if (tid + u * NT < A16) dst[tid + u * NT] = ra[u];
  };
  auto issue_X = [&](int chunk) __attribute__((always_inline)) {
    const bool second = chunk >= a.ncb;
    const float* srcT = second ? a.in2 : a.in;
    const int KK = second ? a.K2 : a.K;
    const int cb = second ? chunk - a.ncb : chunk;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int c = cb * CB + wave * XR + j;
      const float* rowp = srcT + ((long)n * KK + min(c, KK - 1)) * P + (long)t0 * V;
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        rx[j][u] = rowp[(q < xlen) ? q : 0];
      }
    }
  };
  auto commit_X = [&](int chunk) __attribute__((always_inline)) {
    const bool second = chunk >= a.ncb;
    const int KK = second ? a.K2 : a.K;
    const int cb = second ? chunk - a.ncb : chunk;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int cl = wave * XR + j;
      const bool rok = (cb * CB + cl) < KK;
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        if (q < FT * V) xb[cl * XP + q] = (rok && q < xlen) ? rx[j][u] : 0.f;
      }
    }
  };

  // prologue: stage 0 (and 1) in place before the loop
  issue_A(0);
  issue_X(0);
  commit_A(0);
  commit_X(0);
  if (S > 1) issue_A(1);
  __syncthreads();

  float xo[BCH ? 1 : VS];
  bf16x8 xq[2][PL];                                    // split chain: this lane's x fragments (ks, plane)
  // one 16-deep step on the split operands, smallest products first (npl == 1, AGCN_GEMM=bf16: the hi*hi product only)
  auto mfma_step = [&](const bf16x8 (&x)[PL], const bf16x8 (&y)[PL], f32x16 c) __attribute__((always_inline)) {
    if constexpr (F16) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[1]), __builtin_bit_cast(f16x8, y[0]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[1]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[0]), c, 0, 0, 0);
    } else {
      if (a.npl == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], c, 0, 0, 0);
    }
    return c;
  };
  // 8 consecutive fp32 values -> the PL planes of one fragment (range-scaled when F16)
  auto split8 = [&](const float (&v)[8], bf16x8 (&q)[PL], float sc) __attribute__((always_inline)) {
    u32x4 w[3];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      unsigned p0, p1, p2 = 0;
      if constexpr (F16) split_pair_f16(v[2 * e2] * sc, v[2 * e2 + 1] * sc, p0, p1);
      else split_pair(v[2 * e2], v[2 * e2 + 1], p0, p1, p2);
      w[0][e2] = p0; w[1][e2] = p1; w[2][e2] = p2;
    }
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) q[pl] = __builtin_bit_cast(bf16x8, w[pl]);
  };
  long long tk0 = 0, tk_stage = 0, tk_mat = 0, tk_bar = 0;   // AGCN_GC_DBG & 8: cycles of wave 0 per section
  const long long tk_begin = (a.dbg & 8) ? clock64() : 0;
  for (int s = 0; s < S; ++s) {
    const bool plain = s >= S1;
    const int cb = plain ? 0 : s / 3, i = plain ? 0 : s - cb * 3;
    f32x16 d;
    if (a.dbg & 8) tk0 = clock64();
    if (plain) {
      // Plain stage: the B operand is the staged chunk itself (register j of lane (h, v) = channel c_j + 4h), no
      // aggregation.  Every wave takes its 16 values first; only then may the next chunk overwrite the buffer.
      const int chunk = a.ncb + (s - S1);
      const float* xr = xb + wave * V + min(lr, V - 1);
#pragma unroll
      for (int j = 0; j < 16; ++j) d[j] = xr[((j & 3) + 8 * (j >> 2) + 4 * h) * XP];
      if constexpr (F16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] *= rs_s;
      }
      __syncthreads();
      if (chunk + 1 < nchunks) commit_X(chunk + 1);
      if (s + 1 < S) commit_A(s + 1);
      if (s + 2 < S) issue_A(s + 2);
      if (chunk + 2 < nchunks) issue_X(chunk + 2);
    } else if (!(a.dbg & 2)) {
      if (s + 1 < S) commit_A(s + 1);                    // loads were issued one stage ago
      if (i == 2 && cb + 1 < nchunks) commit_X(cb + 1);  // xb is free: every wave took its operands at i == 0
      if (s + 2 < S) issue_A(s + 2);
      if (i == 0 && cb + 1 < nchunks) issue_X(cb + 1);
      if (i == 2 && cb + 1 == a.ncb && cb + 2 < nchunks) issue_X(cb + 2);   // second plain chunk: one stage ahead
    }
    if (a.dbg & 8) { const long long t = clock64(); tk_stage += t - tk0; tk0 = t; }
    if (fvalid && !(a.dbg & 1)) {
      if (!plain) {
        // ---- 1. G = X[cb] . A^_i for this wave's frame ----
        const float* xr = xb + lr * XP + wave * V;
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = 0.f;
        if constexpr (!BCH) {
          const float* ar = adjp + i * 1024 + lr;
          if (i == 0) {
#pragma unroll
            for (int k = 0; k < VS; ++k) xo[k] = xr[2 * k + h];   // u = V (odd V): a finite neighbour or the zeroed pad, times 0
          }
          float bo[BCH ? 1 : VS];
#pragma unroll
          for (int k = 0; k < VS; ++k) bo[k] = ar[(2 * k + h) * 32];
#pragma unroll
          for (int k = 0; k < VS; ++k) d = mfma32(xo[k], bo[k], d);
        } else {
          if (i == 0) {
            // joints 16 ks + 8 h + e of channel lr; beyond V: finite neighbours (next frame / row / zeroed slack) that
            // meet the zero rows of the adjacency
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              float xv[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] = xr[16 * ks + 8 * h + e];
              split8(xv, xq[ks], rs_s);
            }
          }
          const unsigned char* aq = adjq + i * PL * 2048 + (h * 32 + lr) * 16;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bq[PL];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) bq[pl] = *reinterpret_cast<const bf16x8*>(aq + pl * 2048 + ks * 1024);
            d = mfma_step(xq[ks], bq, d);
          }
        }
      }
      // ---- 2. split G in registers and project: acc[tm] += W_i[:, cb] . G ----
      const unsigned char* ab = abuf + (s & 1) * A_IMG + lane * 16;
      bf16x8 gb[2][PL];                                 // (G carries the range scale already)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = d[8 * hf + e];
        split8(gv, gb[hf], 1.f);
      }
      // 2*TM steps of 6 MFMAs; the weight fragments of step k+1 are read while step k runs (order pinned below: the
      // scheduler otherwise sinks every read to its use and waits for it)
      auto load_w = [&](bf16x8 (&af)[PL], int step) __attribute__((always_inline)) {
        const int hf = step / TM, tm = step - hf * TM;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
          af[pl] = *reinterpret_cast<const bf16x8*>(ab + ((pl * 2 + hf) * TM + tm) * 1024);
      };
      bf16x8 afc[PL];
      load_w(afc, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, PL, 0);
#pragma unroll
      for (int step = 0; step < 2 * TM; ++step) {
        const int hf = step / TM, tm = step - hf * TM;
        bf16x8 afn[PL];
        if (step + 1 < 2 * TM) load_w(afn, step + 1);
        acc[tm] = mfma_step(afc, gb[hf], acc[tm]);
        if (step + 1 < 2 * TM) {
          __builtin_amdgcn_sched_group_barrier(0x100, PL, 0);
#pragma unroll
          for (int pl = 0; pl < PL; ++pl) afc[pl] = afn[pl];
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NPROD, 0);
      }
    }
    if (a.dbg & 8) { const long long t = clock64(); tk_mat += t - tk0; tk0 = t; }
    __syncthreads();
    if (a.dbg & 8) { const long long t = clock64(); tk_bar += t - tk0; }
  }
  const long long tk_loop_end = (a.dbg & 8) ? clock64() : 0;

  // ---- epilogue: the block's (BM x FT*V) tile goes through LDS so that every row is stored (and its residual /
  // accumulate operands loaded) as one contiguous run; a row belongs to one wave, which also reduces its
  // (sum, sumsq) partials ----
  // Row groups are software-pipelined: the residual / accumulate operands of group k+1 are loaded RAW into registers
  // (nothing consumes them yet) while group k is combined and stored, so their latency overlaps the stores.
  constexpr int RG = 4, NG = BM / (NW * RG);
  const bool has_extra = a.accumulate || a.add1 || a.add2;   // kernel-uniform
  float ex[2][5][RG][XB];                              // [buffer][out, add1, mask1, add2, mask2]
  auto row_base = [&](int k, int g) __attribute__((always_inline)) {
    const int m = min(m0 + (k * NW + wave) * RG + g, a.M - 1);
    return (((long)n * a.M + m) * T + t0) * V;
  };
  auto load_extras = [&](int k, float (&e)[5][RG][XB]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const long base = row_base(k, g);
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        const long idx = base + ((q < xlen) ? q : 0);
        if (a.accumulate) e[0][g][u] = a.out[idx];
        if (a.add1) e[1][g][u] = a.add1[idx];
        if (a.mask1) e[2][g][u] = mask_load(a.mask1, idx, a.mask_bits);
        if (a.add2) e[3][g][u] = a.add2[idx];
        if (a.mask2) e[4][g][u] = mask_load(a.mask2, idx, a.mask_bits);
      }
    }
  };
  if (has_extra) load_extras(0, ex[0]);                // in flight across the tile transposition below
  float* tile = reinterpret_cast<float*>(smem);        // [BM][TP]; the loop's final barrier freed all of LDS
  const int TP = (FT * V) | 1;                         // odd pitch: column walks over consecutive rows are conflict-free
  if (lr < V) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 16; ++j)
        tile[(tm * 32 + mfma_row(j, h)) * TP + wave * V + lr] = F16 ? acc[tm][j] * rs_inv : acc[tm][j];
  }
  __syncthreads();
  if (a.stats) {
    // per-channel (sum, sumsq) of y = tile + bias over the tile's valid positions: thread <-> (row, column slice)
    constexpr int NP = NT / BM;
    float* red = tile + BM * TP;                       // [NP][2][BM]
    const int r = tid % BM, part = tid / BM;
    const float bval = bias_s[r];
    float bsum = 0.f, bsq = 0.f;
#pragma unroll 8
    for (int q = part; q < xlen; q += NP) {
      const float y = tile[r * TP + q] + bval;
      bsum += y;
      bsq += y * y;
    }
    red[(part * 2 + 0) * BM + r] = bsum;
    red[(part * 2 + 1) * BM + r] = bsq;
    __syncthreads();
    const long slot = (long)n * a.ntiles + tile_id;
    for (int e = tid; e < 2 * BM; e += NT) {
      const int k = e / BM, ml = e - k * BM;
      float sum = 0.f;
#pragma unroll
      for (int p2 = 0; p2 < NP; ++p2) sum += red[(p2 * 2 + k) * BM + ml];
      if (m0 + ml < a.M) a.stats[(slot * 2 + k) * a.M + m0 + ml] = sum;
    }
  }
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int r0 = (k * NW + wave) * RG;
    if (m0 + r0 >= a.M) break;                         // wave-uniform
    if (has_extra && k + 1 < NG) load_extras(k + 1, ex[(k + 1) & 1]);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const int m = m0 + r0 + g;
      if (m >= a.M) break;                             // wave-uniform
      const long base = row_base(k, g);
      const float bval = bias_s[r0 + g];
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        float v = tile[(r0 + g) * TP + min(q, TP - 1)] + bval;
        if (a.accumulate) v += ex[k & 1][0][g][u];
        const long midx = base + ((q < xlen) ? q : 0);
        if (a.add1) v += (!a.mask1 || mask_pass(ex[k & 1][2][g][u], midx, a.mask_bits)) ? ex[k & 1][1][g][u] : 0.f;
        if (a.add2) v += (!a.mask2 || mask_pass(ex[k & 1][4][g][u], midx, a.mask_bits)) ? ex[k & 1][3][g][u] : 0.f;
        if (a.relu) v = fmaxf(v, 0.f);
        if (q < xlen) a.out[base + q] = v;
      }
    }
  }
  if ((a.dbg & 8) && blockIdx.x == gridDim.x / 2 && tid == 0) {
    const long long tk_end = clock64();
    float* o = a.out + ((long)n * a.M + m0) * P + (long)t0 * V;   // (debug run: overwrites 8 outputs of this tile)
    o[0] = (float)(tk_loop_end - tk_begin); o[1] = (float)tk_stage; o[2] = (float)tk_mat; o[3] = (float)tk_bar;
    o[4] = (float)(tk_end - tk_loop_end); o[5] = (float)S;
    o[6] = -12345.f;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-stationary, persistent variant of the chain (f16x3 only) for K <= 64 streamed channels (round 3).
//
// The tile-per-workgroup kernel above spends 90 % of a 64-row workgroup's life outside the matrix pipe: every 4-frame tile
// pays the adjacency conversion, the staging of its 6 weight images through registers, 6 barriers, and an epilogue nothing
// overlaps.  Here ONE workgroup per CU (8 waves = 8 frames per tile) keeps
//   * all weight images of its 64-row block (3*K*64 two-plane fp16 = 48 KB at K = 64, + 8 KB per plain stage) and
//   * the sample's three adjacencies (fp16 planes, 12 KB)
// in LDS for its whole life and walks the frames [f0, f1) of its sample.  Per tile and wave (one frame):
//   * the x fragments (lane = channel, 8 consecutive joints) come STRAIGHT from global memory / L2 into registers, 16-byte
//     buffer loads issued one tile ahead (range-checked: rows past the sample read as 0) -- no x tile in LDS, no barrier on
//     the input side;
//   * 6 (+ plain) stages of  aggregate (6 MFMAs) -> split G in registers -> project (6*TM MFMAs), straight-line;
//   * the accumulators go through ONE LDS tile O [rows][8 frames * V] (two barriers per tile); its rows are stored (bias,
//     residual / accumulate operands, ReLU, BatchNorm partials) as whole 16-byte granules WHILE the next tile's matrix work
//     runs: the row-store of tile k is cut into pieces that are interleaved with the stages of tile k+1.
// BatchNorm partials stay in registers across tiles (a thread always owns the same granules of the tile) and are combined
// once per workgroup in a fixed order: one stats slot per (sample, frame split).
// ------------------------------------------------------------------------------------------------------------------
constexpr int WS_NW = 8;                  // matrix waves = frames per tile
constexpr int WS_SW = 4;                  // store waves
constexpr int WS_NT = (WS_NW + WS_SW) * 64, WS_ST = WS_SW * 64;

struct WsGeom {
  int nmb, nsplit, fper, OPf, ngran;      // row blocks, frame splits per sample, frames per split, O pitch (floats), granules
  unsigned gpr_magic;                     // ceil(2^32 / (OPf / 4)): granule index -> row by one multiply-high
  int off_adj, off_o;                     // LDS byte offsets (weights at 0)
  int o_tiles;                            // 2: O double-buffered (the store waves read tile k-1 while tile k is written)
  size_t smem_bytes, img_bytes;
};

template <int TM>
static inline WsGeom ws_geometry(int N, int M, int K, int K2, int T, int V) {
  constexpr int BM = TM * 32;
  WsGeom g;
  g.nmb = (M + BM - 1) / BM;
  const int ntile = (T + WS_NW - 1) / WS_NW;
  long want = (256 + (long)N * g.nmb - 1) / ((long)N * g.nmb);       // workgroups ~ one per CU
  if (const char* e = getenv("AGCN_WS_SPLIT")) want = atoi(e);     // test knob: frame splits per sample (read per call)
  if (want < 1) want = 1;
  if (want > ntile) want = ntile;
  const int tps = (int)((ntile + want - 1) / want);                   // tiles per split
  g.fper = tps * WS_NW;
  g.nsplit = (ntile + tps - 1) / tps;
  g.OPf = (WS_NW * V + 3) & ~3;
  g.ngran = BM * (g.OPf / 4);
  g.gpr_magic = (unsigned)((0x100000000ull + (unsigned)(g.OPf / 4) - 1) / (unsigned)(g.OPf / 4));
  const int kcb = (K <= 64) ? 2 : 4;                               // channel blocks of the kernel instantiation (ws_dispatch)
  const int stages = 3 * kcb + (K2 + CB - 1) / CB;
  const size_t a_img = (size_t)2 * 2 * TM * 1024;
  g.img_bytes = (size_t)stages * a_img;
  g.off_adj = (int)g.img_bytes;
  g.off_o = g.off_adj + 3 * 2 * 2 * 2 * 32 * 16;
  const size_t o_bytes = (size_t)BM * g.OPf * 4;
  g.o_tiles = (K2 == 0 && (size_t)g.off_o + 2 * o_bytes <= 160 * 1024) ? 2 : 1;   // double-buffered O: one barrier per tile
  g.smem_bytes = (size_t)g.off_o + g.o_tiles * o_bytes;
  return g;
}

// Roles: waves 0..7 = matrix waves (one frame each), waves 8..11 = store waves (the O tile of the PREVIOUS tile -> HBM while
// the matrix waves work on the current one).  Two barriers per tile, executed by both roles:
//     matrix: [stages of tile k] C [accumulators -> O] E            store: [rows of tile k-1] C  E
// C = every store wave is done reading O, E = O of tile k is complete.
// ROLES = (NCB2 == 0), the forward: separate store waves (12 waves, 168 registers each).  With plain stages (the backward-
// data call with its fused 1x1 term: HBM-bound, it moves 1.35 GB at the 64-channel layers against 33 GFLOP) the eight
// matrix waves keep 256 registers and store the previous tile's rows themselves, one piece of granules after every stage.
// EXTRAS: the row-store takes accumulate / residual / mask operands or a ReLU (backward-data, inference); the training
// forward's store is bias + BatchNorm partials only and compiles to a fraction of the code (the general store, unrolled
// over a thread's granules, is tens of KB of instructions: it thrashed the instruction cache of the plain forward).
template <int TM, int KCB, int NCB2, bool EXTRAS>
__global__ void __launch_bounds__(NCB2 == 0 ? WS_NT : WS_NW * 64, NCB2 == 0 ? 3 : 2)
gcn_ws_kernel(const ChainArgs a, const WsGeom g) {
  constexpr bool ROLES = (NCB2 == 0);
  constexpr int NTHREADS = ROLES ? WS_NT : WS_NW * 64;      // threads of the workgroup
  constexpr int STT = ROLES ? WS_ST : WS_NW * 64;           // threads that store rows
  constexpr int BM = TM * 32;
  constexpr int A_IMG = 2 * 2 * TM * 1024;            // bytes of one stage's two-plane weight image
  constexpr int S1 = 3 * KCB, S = S1 + NCB2;
  constexpr bool STATS = (NCB2 == 0);                 // BatchNorm partials: the training forward only (registers)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wimg = smem;
  unsigned char* adjq = smem + g.off_adj;             // [i][plane][ks][h][v][8 u] fp16
  float* O0 = reinterpret_cast<float*>(smem + g.off_o);
  const int o_stride = (g.o_tiles == 2) ? BM * g.OPf : 0;     // floats between the two O tiles (0: single buffer)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int V = a.V, T = a.T;
  const long P = (long)T * V;
  const int bid = blockIdx.x;
  const unsigned long long tk_entry = (a.dbg & 8) ? wall_clock64() : 0;     // AGCN_WS_DBG & 8: see the end of the kernel
  const int split = bid % g.nsplit;
  const int mbk = (bid / g.nsplit) % g.nmb;
  const int n = bid / (g.nsplit * g.nmb);
  const int m0 = mbk * BM;
  const int f0 = split * g.fper, f1 = min(T, f0 + g.fper);
  const int OPf = g.OPf, GPR = OPf >> 2;

  // (the matrix waves' first fragments are issued before everything else: see below)
  // ---- resident state: weight images of this row block, the sample's adjacencies ----
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wp) + (long)mbk * S * (A_IMG / 16);
    u32x4* dst = reinterpret_cast<u32x4*>(wimg);
    for (int e = tid; e < S * (A_IMG / 16); e += NTHREADS) dst[e] = src[e];
    const float* adjn = a.adj + (long)n * 3 * V * V;
    for (int e = tid; e < 3 * 2 * 2 * 32 * 4; e += NTHREADS) {         // one pair (u, u+1) per iteration
      const int e2 = e & 3, col = (e >> 2) & 31, hh = (e >> 7) & 1, ks = (e >> 8) & 1, i = e >> 9;
      float v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int u = 16 * ks + 8 * hh + 2 * e2 + q;
        const bool ok = u < V && col < V;
        const int gi = ok ? (a.adj_t ? ((i * V + col) * V + u) : ((i * V + u) * V + col)) : 0;
        const float tv = adjn[gi];
        v[q] = ok ? tv : 0.f;
      }
      unsigned p0, p1;
      split_pair_f16(v[0], v[1], p0, p1);
      const int o = (((ks * 2 + hh) * 32) + col) * 16 + e2 * 4;
      *reinterpret_cast<unsigned*>(adjq + (i * 2 + 0) * 2048 + o) = p0;
      *reinterpret_cast<unsigned*>(adjq + (i * 2 + 1) * 2048 + o) = p1;
    }
  }
  const int ntiles = (f1 - f0 + WS_NW - 1) / WS_NW;

  float rs_s = 1.f, rs_inv = 1.f;                     // f16x3 range scale of the streamed operands and its inverse
  {
    float mx = a.in_absmax ? *a.in_absmax : 0.f;
    if (a.in2_absmax) mx = fmaxf(mx, *a.in2_absmax);
    f16_range_scale_of<F16_ADJ_TARGET>(mx, rs_s, rs_inv);
    rs_inv *= F16_W_INV;                               // (the packed weights carry F16_W_SCALE)
  }
  // ---- row-store machinery (store waves when ROLES, else the matrix waves themselves) ----
  // A store wave owns whole rows of the O tile: row r = sw + NSW * u, lane <-> 16-byte granule (floats 4 lane .. 4 lane + 3)
  // of the row, so one wave-instruction moves one contiguous run of a row (800 bytes at V = 25), the row offset is a scalar
  // (soffset of the buffer instruction), the bias and every bounds test but "lane < granules per row" are wave-uniform, and
  // the BatchNorm partials of a row live in one register pair per (lane, row) that is summed across lanes once at the end.
  constexpr int NSW = ROLES ? WS_SW : WS_NW;          // waves that store rows
  constexpr int RPW = BM / NSW;                       // rows per store wave
  constexpr int RBW = ROLES ? 4 : 2;                  // rows per batch (their loads are all issued before the first use;
  constexpr int RB = RPW < RBW ? RPW : RBW;           //  the matrix waves of !ROLES have few registers to spare)
  const int sw = ROLES ? wave - WS_NW : wave;         // (negative in the matrix waves of ROLES: unused there)
  float st_s[STATS ? RPW : 1], st_q[STATS ? RPW : 1];
#pragma unroll
  for (int u = 0; u < (STATS ? RPW : 1); ++u) { st_s[u] = 0.f; st_q[u] = 0.f; }
  const bool al16 = ((P & 3) == 0);                   // rows of the output are 16-byte aligned (f0 is a multiple of 8)
  // every operand of the row-store is addressed through a range-checked buffer descriptor over THIS row block of the
  // sample (rows m0 .. m0 + BM - 1, clipped to M: loads past it give 0, stores are dropped)
  const int rows_ok = min(BM, a.M - m0);
  const long blk0 = ((long)n * a.M + m0) * P;         // first element of the row block in the (N, M, T, V) tensors
  const int blk_bytes = (int)((long)rows_ok * P * 4);
  auto mk = [&](const void* p, int bytes) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? bytes : 0, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t r_out = mk(a.out + blk0, blk_bytes);
  const __amdgpu_buffer_rsrc_t r_a1 = mk(a.add1 ? a.add1 + blk0 : nullptr, blk_bytes);
  const __amdgpu_buffer_rsrc_t r_a2 = mk(a.add2 ? a.add2 + blk0 : nullptr, blk_bytes);
  // masks: fp32 tensors like the operands, or sign-bit words (bit e of word w <-> element 32 w + e)
  const int sh0 = (int)(blk0 & 31);
  const int mbytes = a.mask_bits ? (int)((((long)rows_ok * P + sh0 + 31) >> 5) * 4) : blk_bytes;
  const __amdgpu_buffer_rsrc_t r_m1 =
      mk(a.mask1 ? (a.mask_bits ? (const void*)(reinterpret_cast<const unsigned*>(a.mask1) + (blk0 >> 5)) : (const void*)(a.mask1 + blk0)) : nullptr, mbytes);
  const __amdgpu_buffer_rsrc_t r_m2 =
      mk(a.mask2 ? (a.mask_bits ? (const void*)(reinterpret_cast<const unsigned*>(a.mask2) + (blk0 >> 5)) : (const void*)(a.mask2 + blk0)) : nullptr, mbytes);
  // element offset eo (inside the row block) -> 16 bytes / one float / ReLU-mask bits
  auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) __attribute__((always_inline)) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  };
  auto ld1 = [&](const __amdgpu_buffer_rsrc_t& r, int eo) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, eo * 4, 0, 0));
  };
  auto mask4 = [&](const __amdgpu_buffer_rsrc_t& r, int eo) __attribute__((always_inline)) {   // eo a multiple of 4
    if (a.mask_bits) {
      const int b = sh0 + eo;
      return (__builtin_amdgcn_raw_buffer_load_b32(r, (b >> 5) * 4, 0, 0) >> (b & 31)) & 15u;
    }
    const f32x4 m = ld4(r, eo * 4, 0);
    return (m[0] > 0.f ? 1u : 0u) | (m[1] > 0.f ? 2u : 0u) | (m[2] > 0.f ? 4u : 0u) | (m[3] > 0.f ? 8u : 0u);
  };
  auto mask1b = [&](const __amdgpu_buffer_rsrc_t& r, int eo) __attribute__((always_inline)) {
    if (a.mask_bits) {
      const int b = sh0 + eo;
      return ((__builtin_amdgcn_raw_buffer_load_b32(r, (b >> 5) * 4, 0, 0) >> (b & 31)) & 1u) != 0u;
    }
    return ld1(r, eo) > 0.f;
  };
  struct Extras { f32x4 o, acc, a1, a2; unsigned m1, m2; };
  const int q = lane * 4;                             // first float of this lane's granule
  float bias_r[RPW];                                  // bias of this wave's rows (uniform values, loaded once)
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int r = sw + NSW * u;
    bias_r[u] = (a.bias && r >= 0 && r < rows_ok) ? a.bias[m0 + r] : 0.f;
  }
  // rows [b0, b0 + RB) of this wave, frames t0s .. of the tile, `nvalid` valid floats per row (a multiple of 4 here)
  auto store_rows_fast = [&](const float* O, int b0, int t0s, int nvalid) __attribute__((always_inline)) {
    Extras ex[RB];
    const bool on = q < nvalid;                        // (lanes beyond the row idle)
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int r = sw + NSW * (b0 + u);               // wave-uniform
      if (r >= rows_ok || !on) continue;
      ex[u].o = *reinterpret_cast<const f32x4*>(O + r * OPf + q);
      if constexpr (EXTRAS) {
        const int ro = (r * T + t0s) * V;              // scalar: first element of the row's run
        ex[u].m1 = ex[u].m2 = 15u;
        if (a.accumulate) ex[u].acc = ld4(r_out, q * 4, ro * 4);
        if (a.add1) { ex[u].a1 = ld4(r_a1, q * 4, ro * 4); if (a.mask1) ex[u].m1 = mask4(r_m1, ro + q); }
        if (a.add2) { ex[u].a2 = ld4(r_a2, q * 4, ro * 4); if (a.mask2) ex[u].m2 = mask4(r_m2, ro + q); }
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int r = sw + NSW * (b0 + u);
      if (r >= rows_ok || !on || (a.dbg & 4)) continue;
      const float bval = bias_r[b0 + u];
      f32x4 v = ex[u].o;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * rs_inv + bval;      // (the accumulators carry the range scale)
      if constexpr (STATS) {
        if (a.stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { st_s[b0 + u] += v[e]; st_q[b0 + u] += v[e] * v[e]; }
        }
      }
      if constexpr (EXTRAS) {
        if (a.accumulate) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += ex[u].acc[e];
        }
        if (a.add1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += ((ex[u].m1 >> e) & 1u) ? ex[u].a1[e] : 0.f;
        }
        if (a.add2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += ((ex[u].m2 >> e) & 1u) ? ex[u].a2[e] : 0.f;
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, q * 4, (r * T + t0s) * V * 4, 0);
    }
  };
  // rows that are not 16-byte aligned (T*V not a multiple of 4) or a partial tile whose run is not a multiple of 4 floats:
  // element by element, deliberately compact code (rare: odd shapes and at most the last tile of a frame range)
  auto store_rows_slow = [&](const float* O, int b0, int t0s, int nvalid) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int r = sw + NSW * (b0 + u);
      if (r >= rows_ok || (a.dbg & 4)) continue;
      const float bval = bias_r[b0 + u];
      const int ro = (r * T + t0s) * V;
      float ss = 0.f, sq = 0.f;
#pragma unroll 1
      for (int e = 0; e < 4; ++e) {
        if (q + e >= nvalid) break;
        const float z = O[r * OPf + q + e] * rs_inv + bval;
        ss += z;
        sq += z * z;
        float y = z;
        if constexpr (EXTRAS) {
          const int eo = ro + q + e;
          if (a.accumulate) y += ld1(r_out, eo);
          if (a.add1) y += (!a.mask1 || mask1b(r_m1, eo)) ? ld1(r_a1, eo) : 0.f;
          if (a.add2) y += (!a.mask2 || mask1b(r_m2, eo)) ? ld1(r_a2, eo) : 0.f;
          if (a.relu) y = fmaxf(y, 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), r_out, (ro + q + e) * 4, 0, 0);
      }
      if constexpr (STATS) {
        if (a.stats) { st_s[b0 + u] += ss; st_q[b0 + u] += sq; }
      }
    }
  };
  auto store_tile = [&](const float* O, int t0s, int nvalid) __attribute__((always_inline)) {
    const bool fast = al16 && (nvalid & 3) == 0;       // wave-uniform
#pragma unroll
    for (int b0 = 0; b0 < RPW; b0 += RB) {
      if (fast) store_rows_fast(O, b0, t0s, nvalid);
      else store_rows_slow(O, b0, t0s, nvalid);
    }
  };
  // BatchNorm partials of this wave's rows: lanes summed in a fixed butterfly order, one slot per (sample, frame split)
  auto store_stats = [&]() __attribute__((always_inline)) {
    if constexpr (STATS) {
      if (!a.stats) return;
      const long slot = (long)n * g.nsplit + split;
#pragma unroll
      for (int u = 0; u < RPW; ++u) {
        float s1 = st_s[u], s2 = st_q[u];
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) { s1 += __shfl_xor(s1, k); s2 += __shfl_xor(s2, k); }
        const int r = sw + NSW * u;
        if (lane == 0 && r < rows_ok) {
          a.stats[(slot * 2 + 0) * a.M + m0 + r] = s1;
          a.stats[(slot * 2 + 1) * a.M + m0 + r] = s2;
        }
      }
    }
  };

  if (ROLES && wave >= WS_NW) {
    // =================================== store role ===================================
    __builtin_amdgcn_s_setprio(3);                      // few, latency-bound instructions: ahead of the matrix waves' stream
    __syncthreads();                                    // (resident images in place: the matrix waves start)
    const unsigned long long tk_pro = (a.dbg & 8) ? wall_clock64() : 0;
    unsigned long long tk_loop = 0;
    for (int k = 0; k <= ntiles; ++k) {                 // (one call site of the store code: instruction-cache footprint)
      if (k == ntiles && (a.dbg & 8)) tk_loop = wall_clock64();
      if (k > 0) {
        const int t0s = f0 + (k - 1) * WS_NW;
        store_tile(O0 + ((k - 1) & 1) * o_stride, t0s, min(WS_NW, f1 - t0s) * V);
      }
      if (k == ntiles) break;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // C: done reading O (tile k-1)
      if (o_stride == 0) asm volatile("s_barrier" ::: "memory");        // E: O of tile k is complete (single buffer)
    }
    store_stats();
    if ((a.dbg & 8) && tid == WS_NW * 64) {             // (debug run: overwrites 4 outputs of this workgroup's block)
      float* o = a.out + blk0 + (long)f0 * V + 16;
      o[0] = (float)(tk_pro - tk_entry); o[1] = (float)(tk_loop - tk_pro); o[2] = (float)(wall_clock64() - tk_loop);
      o[3] = -54321.f;
    }
    return;
  }

  // =================================== matrix role ===================================
  // ---- streamed operands: range-checked buffer loads (a row past the sample reads as 0) ----
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (long)n * a.K * P), 0, (int)((long)a.K * P * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(NCB2 ? a.in2 + (long)n * a.K2 * P : a.in), 0, NCB2 ? (int)((long)a.K2 * P * 4) : 0, 0x00020000);
  // x fragment of channel block cb, k-step ks: joints 16 ks + 8 h + [0, 8) of channel 32 cb + lr, frame t
  int xoff[KCB];
#pragma unroll
  for (int cb = 0; cb < KCB; ++cb) xoff[cb] = (int)(((long)min(cb * CB + lr, a.K - 1) * P + 8 * h) * 4);
  auto load_x = [&](int t, int cb, f32x4 (&xc)[4]) __attribute__((always_inline)) {
    const int tb = t * V * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // q = 2 ks + half: floats 16 ks + 4 (q & 1)
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[cb] + tb + ((q >> 1) * 16 + (q & 1) * 4) * 4, 0, 0);
      xc[q] = __builtin_bit_cast(f32x4, r);
    }
  };
  // plain-stage operand of 32-channel block s2 at frame t (16 scalar loads: lanes = joints, coalesced 100-byte runs)
  auto load_p = [&](int s2, int t, float (&dv)[16]) __attribute__((always_inline)) {
    const int pos = (t * V + min(lr, V - 1)) * 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c2 = min(s2 * CB + (j & 3) + 8 * (j >> 2) + 4 * h, a.K2 - 1);
      dv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx2, (int)((long)c2 * P * 4) + pos, 0, 0));
    }
  };
  // one 16-deep step of the three fp16 products (smallest first)
  auto mfma3 = [&](const bf16x8 (&x)[2], const bf16x8 (&y)[2], f32x16 c) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[1]), __builtin_bit_cast(f16x8, y[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[0]), c, 0, 0, 0);
    return c;
  };
  auto split8 = [&](const float (&v)[8], bf16x8 (&q)[2], float sc) __attribute__((always_inline)) {
    u32x4 w0, w1;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      unsigned p0, p1;
      split_pair_f16_mix(v[2 * e2] * sc, v[2 * e2 + 1] * sc, p0, p1);
      w0[e2] = p0; w1[e2] = p1;
    }
    q[0] = __builtin_bit_cast(bf16x8, w0);
    q[1] = __builtin_bit_cast(bf16x8, w1);
  };
  auto splitx = [&](const f32x4 (&xc)[4], bf16x8 (&xq)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float xv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[e] = xc[2 * ks][e]; xv[4 + e] = xc[2 * ks + 1][e]; }
      split8(xv, xq[ks], rs_s);
    }
  };
  // G = X[cb] . A^_i for this wave's frame (6 MFMAs)
  auto aggregate = [&](const bf16x8 (&xq)[2][2], int i) __attribute__((always_inline)) {
    f32x16 d;
#pragma unroll
    for (int j = 0; j < 16; ++j) d[j] = 0.f;
    const unsigned char* aq = adjq + i * 2 * 2048 + (h * 32 + lr) * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 bq[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bq[pl] = *reinterpret_cast<const bf16x8*>(aq + pl * 2048 + ks * 1024);
      d = mfma3(xq[ks], bq, d);
    }
    return d;
  };
  // the 16 values of a lane (one 32-channel block, this lane's joint) -> the two k-steps of the projection's B operand
  auto gsplit = [&](const float (&dv)[16], bf16x8 (&gb)[2][2], float sc) __attribute__((always_inline)) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float gv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = dv[8 * hf + e];
      split8(gv, gb[hf], sc);
    }
  };
  auto project = [&](int stage, const bf16x8 (&gb)[2][2], f32x16 (&acc)[TM]) __attribute__((always_inline)) {
    const unsigned char* ab = wimg + stage * A_IMG + lane * 16;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        bf16x8 af[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(ab + ((pl * 2 + hf) * TM + tm) * 1024);
        acc[tm] = mfma3(af, gb[hf], acc[tm]);
      }
  };

  // ONE fragment register set: channel block 0 of a tile is loaded at the middle of the previous tile (and consumed at the
  // tile's first instruction), blocks 1.. of a tile right after that (consumed two and a half stages later).
  f32x4 xr[4];
  float dvb[16];
  const int t_first = f0 + wave;
  if (t_first < f1) load_x(t_first, 0, xr);
  __syncthreads();                                      // resident images in place
  const unsigned long long tk_pro = (a.dbg & 8) ? wall_clock64() : 0;
  const unsigned long long ck_pro = (a.dbg & 8) ? clock64() : 0;        // shader clock: loop cycles -> in-kernel frequency

  long long ck_st = 0, ck_c = 0, ck_o = 0, ck_e = 0, ck0 = 0;   // (dbg & 8) cycles in stages / barrier C / O write / barrier E
  int prev_t0 = -1;                                     // (!ROLES) tile whose rows are still to be stored
  for (int t0 = f0; t0 < f1; t0 += WS_NW) {
    const int t = t0 + wave;
    const bool fvalid = t < f1 && !(a.dbg & 1);         // wave-uniform
    const bool nvalid_next = (t + WS_NW) < f1 && !(a.dbg & 2);
    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;
    if (a.dbg & 8) ck0 = clock64();
    if (fvalid) {
      // Software pipeline over the stages, interleaved BY HAND (hipcc puts each stage's register split behind its twelve
      // projection MFMAs and reads every LDS fragment right in front of its use; the split's mixed-precision fmas are inline
      // asm, which sched_group_barrier cannot place).  Stage s:
      //     aggregate(s+1): 6 MFMAs on adjacency fragments that were read during stage s-1
      //     2*TM projection steps of stage s: [weight fragments of the NEXT step] [3 MFMAs] [a slice of the split of G(s+1)]
      // with a scheduling fence after every step, so the order written here is the order issued.
      if (NCB2 > 0) load_p(0, t, dvb);                  // first plain operand: in flight during the aggregated stages
      bf16x8 xq[2][2];
      u32x4 gw[2][2][2];                                // [stage parity][k-step half][plane]: B operand of the projection
      bf16x8 bq[2][2];                                  // adjacency fragments [k-step][plane] of the next aggregation
      auto load_adj = [&](int i) __attribute__((always_inline)) {
        const unsigned char* aq = adjq + i * 2 * 2048 + (h * 32 + lr) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) bq[ks][pl] = *reinterpret_cast<const bf16x8*>(aq + pl * 2048 + ks * 1024);
      };
      auto agg = [&]() __attribute__((always_inline)) {
        f32x16 d;
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) d = mfma3(xq[ks], bq[ks], d);
        return d;
      };
      // pairs [p0, p0 + np) of the 8 value pairs of a lane's 16 values -> the two fp16 planes (pair p: half p >> 2, dword p & 3)
      auto split_pairs = [&](const float (&dv)[16], int p0, int np, u32x4 (&gwn)[2][2], float sc) __attribute__((always_inline)) {
#pragma unroll
        for (int p = p0; p < p0 + np; ++p) {
          unsigned w0, w1;
          split_pair_f16_mix(dv[2 * p] * sc, dv[2 * p + 1] * sc, w0, w1);
          gwn[p >> 2][0][p & 3] = w0;
          gwn[p >> 2][1][p & 3] = w1;
        }
      };
      auto load_w = [&](int stage, int st, bf16x8 (&af)[2]) __attribute__((always_inline)) {
        const int hf = st / TM, tm = st - hf * TM;
        const unsigned char* ab = wimg + stage * A_IMG + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(ab + ((pl * 2 + hf) * TM + tm) * 1024);
      };
      constexpr int NST = 2 * TM;                       // projection steps per stage
      // value pairs of G(s+1) split behind step st: none behind step 0 (the aggregation that produces them was issued just
      // before it: its six MFMAs have to drain first), the eight pairs spread over the remaining steps
      constexpr int PP0[4] = {0, 0, 3, 6}, PPN[4] = {0, 3, 3, 2};       // NST == 4
      constexpr int QP0[2] = {0, 0}, QPN[2] = {0, 8};                   // NST == 2
      splitx(xr, xq);
      if (KCB > 1) load_x(t, 1, xr);                    // (the registers just consumed take the next channel block)
      else if (nvalid_next) load_x(t + WS_NW, 0, xr);
      load_adj(0);
      {
        const f32x16 d = agg();
        load_adj(S1 > 1 ? 1 : 0);
        float dv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) dv[j] = d[j];
        split_pairs(dv, 0, 8, gw[0], 1.f);
      }
      bf16x8 afc[2], afn[2];
      load_w(0, 0, afc);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int sn = s + 1;
        float dv[16];
        if (sn < S1) {
          if (sn % 3 == 0) {                            // next channel block: its x fragment
            splitx(xr, xq);
            if (sn / 3 + 1 < KCB) load_x(t, sn / 3 + 1, xr);
            else if (nvalid_next) load_x(t + WS_NW, 0, xr);          // the next tile's first block
          }
          const f32x16 dn = agg();                      // (its adjacency fragments were read during the previous stage)
#pragma unroll
          for (int j = 0; j < 16; ++j) dv[j] = dn[j];
        } else if (sn < S) {
#pragma unroll
          for (int j = 0; j < 16; ++j) dv[j] = dvb[j];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          if (!(a.dbg & 32)) {
            if (st + 1 < NST) load_w(s, st + 1, afn);
            else if (sn < S) load_w(sn, 0, afn);
          }
          if (st == NST - 2 && sn + 1 < S1 && !(a.dbg & 64)) load_adj((sn + 1) % 3);   // fragments of the aggregation after the next one
          __builtin_amdgcn_sched_barrier(0);            // (the LDS reads of the next step go out BEFORE this step's MFMAs)
          {
            const int hf = st / TM, tm = st - hf * TM;
            bf16x8 gbv[2] = {__builtin_bit_cast(bf16x8, gw[s & 1][hf][0]), __builtin_bit_cast(bf16x8, gw[s & 1][hf][1])};
            acc[tm] = mfma3(afc, gbv, acc[tm]);
          }
          if (sn < S && !(a.dbg & 16)) split_pairs(dv, NST == 4 ? PP0[st] : QP0[st], NST == 4 ? PPN[st] : QPN[st], gw[sn & 1], sn < S1 ? 1.f : rs_s);
          afc[0] = afn[0]; afc[1] = afn[1];
          __builtin_amdgcn_sched_barrier(0);
        }
        if (sn >= S1 && sn + 1 < S) load_p(sn + 1 - S1, t, dvb);     // (the plain operand just split: its registers are free)
        if constexpr (!ROLES) {                         // a batch of the previous tile's rows after each of the first stages
          if (prev_t0 >= 0 && s * RB < RPW) {
            if (al16) store_rows_fast(O0, s * RB, prev_t0, WS_NW * V); else store_rows_slow(O0, s * RB, prev_t0, WS_NW * V);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (!ROLES) {
      if (prev_t0 >= 0 && !fvalid) store_tile(O0, prev_t0, WS_NW * V);     // (no stages ran in this wave: all its rows now)
      prev_t0 = t0;
    }
    if (a.dbg & 8) { const long long c = clock64(); ck_st += c - ck0; ck0 = c; }
    // single O tile: barrier C (everyone is done reading the previous tile's O), write, barrier E (O complete).
    // two O tiles (ROLES): write tile k's buffer straight away, ONE barrier (the store waves have finished tile k-1's
    // buffer when they arrive, i.e. before anyone overwrites it at tile k+1)
    if (o_stride == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (a.dbg & 8) { const long long c = clock64(); ck_c += c - ck0; ck0 = c; }
    if (fvalid && lr < V) {
      float* O = O0 + (((t0 - f0) / WS_NW) & 1) * o_stride;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[(tm * 32 + mfma_row(j, h)) * OPf + wave * V + lr] = acc[tm][j];   // (still range-scaled)
    }
    if (a.dbg & 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long c = clock64(); ck_o += c - ck0; ck0 = c; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // E: O of this tile is complete
    if (a.dbg & 8) { const long long c = clock64(); ck_e += c - ck0; ck0 = c; }
  }
  const unsigned long long tk_loop = (a.dbg & 8) ? wall_clock64() : 0;
  if ((a.dbg & 8) && tid == 0) {                        // (debug run: overwrites 4 outputs of this workgroup's block)
    float* o = a.out + blk0 + (long)f0 * V;
    o[0] = (float)(tk_pro - tk_entry); o[1] = (float)(tk_loop - tk_pro); o[2] = (float)(clock64() - ck_pro); o[3] = -12345.f;
    o[4] = (float)ck_st; o[5] = (float)ck_c; o[6] = (float)ck_o; o[7] = (float)ck_e;
  }
  if constexpr (!ROLES) {
    store_tile(O0, prev_t0, min(WS_NW, f1 - prev_t0) * V);  // the last tile's rows
    store_stats();
  }
}

struct ChainGeom {
  int ntiles, ncb, nmb, XP, off_bias;
  size_t smem_bytes, pack_bytes;
};

template <int TM, int NW>
ChainGeom chain_geometry(int V, int T, int M, int K, bool bch = false, int planes = 3) {
  constexpr int BM = TM * 32, FT = NW;
  ChainGeom g;
  g.ntiles = (T + FT - 1) / FT;
  g.ncb = (K + CB - 1) / CB;
  g.nmb = (M + BM - 1) / BM;
  g.XP = (FT * V) | 1;
  const size_t a_img = (size_t)planes * 2 * TM * 1024;
  const size_t xb_bytes = ((size_t)CB * g.XP * 4 + (bch ? 128 : 0) + 15) & ~(size_t)15;
  g.smem_bytes = (bch ? (size_t)3 * planes * 2 * 2 * 32 * 16 : (size_t)3 * 32 * 32 * 4) + xb_bytes + 2 * a_img;
  const size_t epi_bytes = (size_t)BM * ((FT * V) | 1) * 4 + (size_t)NW * 64 * 2 * 4;
  if (epi_bytes > g.smem_bytes) g.smem_bytes = epi_bytes;
  g.smem_bytes = (g.smem_bytes + 15) & ~(size_t)15;
  g.off_bias = (int)g.smem_bytes;
  g.smem_bytes += (size_t)BM * 4;
  g.pack_bytes = (size_t)g.nmb * g.ncb * 3 * a_img;
  return g;
}

struct ChainW2 {
  const float* w;
  long sa_m, sa_c;      // W2[m][k] = w[m*sa_m + k*sa_c]
};

template <int TM, int VS, int NW, bool F16 = false>
int chain_launch(ChainArgs a, const float* w, long sa_m, long sa_i, long sa_c, const ChainW2& g_w2, void* ws,
                 size_t ws_bytes, hipStream_t stream) {
  constexpr int PL = F16 ? 2 : 3;
  const ChainGeom g = chain_geometry<TM, NW>(a.V, a.T, a.M, a.K, VS == 0, PL);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  a.ncb2 = a.in2 ? (a.K2 + CB - 1) / CB : 0;
  const int s_total = 3 * g.ncb + a.ncb2;
  const size_t a_img = (size_t)PL * 2 * TM * 1024;
  const size_t img_bytes = (size_t)g.nmb * s_total * a_img;
  if (img_bytes + (F16 ? 16 : 0) > ws_bytes) return AGCN_ERR_WORKSPACE;
  if constexpr (F16) {
    // maxima the caller did not supply: one streaming pass each, into the 16 bytes behind the weight images (the
    // workspace is sized for three planes, the f16x3 images take two)
    unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + img_bytes);
    if (!a.in_absmax) {
      if (int rc = agcn_launch_absmax(a.in, (long)a.N * a.K * a.T * a.V, amax, stream)) return rc;
      a.in_absmax = reinterpret_cast<const float*>(amax);
    }
    if (a.in2 && !a.in2_absmax) {
      if (int rc = agcn_launch_absmax(a.in2, (long)a.N * a.K2 * a.T * a.V, amax + 1, stream)) return rc;
      a.in2_absmax = reinterpret_cast<const float*>(amax + 1);
    }
    if (!a.in2) a.in2_absmax = nullptr;
  }
  a.ntiles = g.ntiles; a.ncb = g.ncb; a.nmb = g.nmb; a.XP = g.XP; a.off_bias = g.off_bias;
  a.wp = (const unsigned short*)ws;
  ChainPackArgs pk;
  pk.w = w; pk.wp = (unsigned short*)ws; pk.M = a.M; pk.K = a.K; pk.ncb = g.ncb;
  pk.sa_m = sa_m; pk.sa_i = sa_i; pk.sa_c = sa_c;
  pk.nsub = 3; pk.s_total = s_total; pk.s_off = 0;
  hipLaunchKernelGGL((chain_pack_kernel<TM, F16>), dim3(g.nmb * g.ncb * 3), dim3(256), 0, stream, pk);
  int rc = agcn_check_launch();
  if (rc) return rc;
  if (a.ncb2 > 0) {        // images of the plain stages, after the aggregated ones of each row block
    ChainPackArgs p2;
    p2.w = g_w2.w; p2.wp = (unsigned short*)ws; p2.M = a.M; p2.K = a.K2; p2.ncb = a.ncb2;
    p2.sa_m = g_w2.sa_m; p2.sa_i = 0; p2.sa_c = g_w2.sa_c;
    p2.nsub = 1; p2.s_total = s_total; p2.s_off = 3 * g.ncb;
    hipLaunchKernelGGL((chain_pack_kernel<TM, F16>), dim3(g.nmb * a.ncb2), dim3(256), 0, stream, p2);
    rc = agcn_check_launch();
    if (rc) return rc;
  }
  auto kern = gcn_chain_kernel<TM, VS, NW, F16>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};   // per (kernel instantiation, device): the attribute is per device
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * g.ntiles * g.nmb)), dim3(NW * 64), g.smem_bytes, stream, a);
  AGCN_NOTE_KERNEL("gcn_chain_kernel<%d, %d, %d, %s>", TM, VS, NW, F16 ? "true" : "false");
  return agcn_check_launch();
}


// ---- weight-stationary launch (see gcn_ws_kernel) ----
static inline bool ws_enabled() {
  static const int on = getenv("AGCN_CHAIN_WS") ? atoi(getenv("AGCN_CHAIN_WS")) : 1;
  return on != 0;
}
// shapes the persistent kernel takes: f16x3 chain, 33..64 streamed channels with up to three plain 32-channel stages, or
// 65..128 streamed channels without plain stages (96 KB of resident weights per 64-row block, single O tile)
static inline bool ws_shape_ok(int M, int K, int K2, int V) {
  if (!(agcn_chain_f16x3() && ws_enabled() && V <= 32 && M >= 1 && K > 32)) return false;
  static const int k128 = getenv("AGCN_WS_K128") ? atoi(getenv("AGCN_WS_K128")) : 1;
  return (K <= 64 && K2 >= 0 && K2 <= 96) || (k128 && K <= 128 && K2 == 0 && M > 32);
}

template <int TM, int NCB2, int KCB = 2>
int ws_launch(ChainArgs a, const float* w, long sa_m, long sa_i, long sa_c, const ChainW2& g_w2, void* ws, size_t ws_bytes,
              hipStream_t stream) {
  const WsGeom g = ws_geometry<TM>(a.N, a.M, a.K, a.K2, a.T, a.V);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if (NCB2 > 0 && a.stats) return AGCN_ERR_UNSUPPORTED;            // (BatchNorm partials: plain-stage-free forward only)
  const size_t img_total = (size_t)g.nmb * g.img_bytes;
  if (img_total + 16 > ws_bytes) return AGCN_ERR_WORKSPACE;
  unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + img_total);
  if (!a.in_absmax) {
    if (int rc = agcn_launch_absmax(a.in, (long)a.N * a.K * a.T * a.V, amax, stream)) return rc;
    a.in_absmax = reinterpret_cast<const float*>(amax);
  }
  if (a.in2 && !a.in2_absmax) {
    if (int rc = agcn_launch_absmax(a.in2, (long)a.N * a.K2 * a.T * a.V, amax + 1, stream)) return rc;
    a.in2_absmax = reinterpret_cast<const float*>(amax + 1);
  }
  if (!a.in2) a.in2_absmax = nullptr;
  a.ncb = KCB; a.ncb2 = NCB2; a.nmb = g.nmb;
  a.wp = (const unsigned short*)ws;
  const int s_total = 3 * KCB + NCB2;
  ChainPackArgs pk;
  pk.w = w; pk.wp = (unsigned short*)ws; pk.M = a.M; pk.K = a.K; pk.ncb = KCB;
  pk.sa_m = sa_m; pk.sa_i = sa_i; pk.sa_c = sa_c;
  pk.nsub = 3; pk.s_total = s_total; pk.s_off = 0;
  hipLaunchKernelGGL((chain_pack_kernel<TM, true>), dim3(g.nmb * KCB * 3), dim3(256), 0, stream, pk);
  int rc = agcn_check_launch();
  if (rc) return rc;
  if (NCB2 > 0) {
    ChainPackArgs p2;
    p2.w = g_w2.w; p2.wp = (unsigned short*)ws; p2.M = a.M; p2.K = a.K2; p2.ncb = NCB2;
    p2.sa_m = g_w2.sa_m; p2.sa_i = 0; p2.sa_c = g_w2.sa_c;
    p2.nsub = 1; p2.s_total = s_total; p2.s_off = 3 * KCB;
    hipLaunchKernelGGL((chain_pack_kernel<TM, true>), dim3(g.nmb * NCB2), dim3(256), 0, stream, p2);
    rc = agcn_check_launch();
    if (rc) return rc;
  }
  const bool extras = NCB2 > 0 || a.accumulate || a.add1 || a.add2 || a.relu;
  const dim3 grid((unsigned)(a.N * g.nmb * g.nsplit)), block(NCB2 == 0 ? WS_NT : WS_NW * 64);
  if (extras) {
    auto kern = gcn_ws_kernel<TM, KCB, NCB2, true>;
    static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};
    if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
    hipLaunchKernelGGL(kern, grid, block, g.smem_bytes, stream, a, g);
  } else {
    if constexpr (NCB2 == 0) {
      auto kern = gcn_ws_kernel<TM, KCB, 0, false>;
      static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};
      if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
      hipLaunchKernelGGL(kern, grid, block, g.smem_bytes, stream, a, g);
    }
  }
  AGCN_NOTE_KERNEL("gcn_ws_kernel<%d, %d, %d>", TM, KCB, NCB2);
  return agcn_check_launch();
}

template <int TM>
int ws_dispatch(const ChainArgs& a, const float* w, long sa_m, long sa_i, long sa_c, const ChainW2& w2, void* ws,
                size_t ws_bytes, hipStream_t stream) {
  const int ncb2 = a.in2 ? (a.K2 + CB - 1) / CB : 0;
  if (a.K > 64) {
    if constexpr (TM == 2) {
      if (ncb2 == 0) return ws_launch<2, 0, 4>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
    }
    return AGCN_ERR_UNSUPPORTED;
  }
  switch (ncb2) {
    case 0: return ws_launch<TM, 0>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
    case 1: return ws_launch<TM, 1>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
    case 2: return ws_launch<TM, 2>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
    case 3: return ws_launch<TM, 3>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
  }
  return AGCN_ERR_UNSUPPORTED;
}

template <int TM, int NW>
int chain_dispatch_vs(const ChainArgs& a, const float* w, long sa_m, long sa_i, long sa_c, const ChainW2& w2, void* ws,
                      size_t ws_bytes, hipStream_t stream) {
  // split-bf16 aggregation unless AGCN_CHAIN_F32=1 (exact-f32 MFMA chain, VS = ceil(V/2) steps) or AGCN_GEMM=bf16
  // (one product: keep the aggregation exact)
  static const int f32chain = getenv("AGCN_CHAIN_F32") ? atoi(getenv("AGCN_CHAIN_F32")) : 0;
  if (!f32chain && a.npl == 3 && agcn_chain_f16x3())
    return chain_launch<TM, 0, NW, true>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
  if (!f32chain && a.npl == 3) return chain_launch<TM, 0, NW>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
  const int vs = (a.V + 1) / 2;
  if (vs == 13) return chain_launch<TM, 13, NW>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
  if (vs == 9) return chain_launch<TM, 9, NW>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
  return chain_launch<TM, 16, NW>(a, w, sa_m, sa_i, sa_c, w2, ws, ws_bytes, stream);
}


// ------------------------------------------------------------------------------------------------------------------
// Adjacency gradient  dA^_i[u][v] = sum_{c,t} x[c][t,u] * H_i[c][t,v],   H_i = Wd_i^T dy   (reference agcn.py:102-105
// differentiated).  Same wave <-> frame mapping: the projection H (split-bf16 MFMA, K = Cout) leaves
// a 32-channel x V tile per accumulator whose D registers feed the exact-f32 reduction MFMA against x directly
// (k-pair of register j = channels {c_j, c_j + 4}); H never leaves the register file.
// A workgroup owns BM rows (i, c) of one subset and NW frames; it writes one (V x V) partial per (sample, subset, slot).
// ------------------------------------------------------------------------------------------------------------------
struct DadjArgs {
  const float* dy;             // (N, Cout, T, V)
  const unsigned short* wp;    // packed images [mblock][kchunk][plane][ks][tm][lane][8] (bf16)
  const float* x;              // (N, C, T, V)
  float* dpart;                // (N, 3, nslots, V, V)
  int N, C, Cout, T, V;
  int ntiles, nkc, nmb, nslots, gpc;   // gpc = row blocks per subset (C / BM)
  int npl;                     // 3: six split products (bf16x6) ; 1: hi*hi only (bf16)
  const float* dy_absmax;      // f16x3: device scalar max |dy| for the range scale
  const float* x_absmax;       // f16x3 reduction against x (null: exact-f32 reduction MFMA)
  int dbg;                     // AGCN_DADJ_DBG profiling switches (1: no reduction phase, 2: no projection MFMAs)
};

struct DadjPackArgs {
  const float* w;              // wcat (Cout, 3C)
  unsigned short* wp;
  int C3, Cout, nkc;
};

constexpr int KC = 32;         // dy channels per stage (two 16-deep MFMA steps)

// one block per (mblock, kchunk) image: [plane][ks][tm][lane][8]; slot e of lane (row m, h) = channel kc*32 + ks*16 + h*8 + e
template <int TM, bool F16>
__global__ void __launch_bounds__(256) dadj_pack_kernel(const DadjPackArgs p) {
  constexpr int BM = TM * 32;
  constexpr int PL = F16 ? 2 : 3;
  constexpr int PER_PLANE = 2 * TM * 64 * 8;
  const int kc = blockIdx.x % p.nkc, mb = blockIdx.x / p.nkc;
  unsigned short* dst = p.wp + (long)blockIdx.x * PL * PER_PLANE;
  for (int e = threadIdx.x; e < PER_PLANE / 2; e += 256) {
    const int e2 = e & 3;
    const int lane = (e >> 2) & 63;
    const int r = e >> 8;
    const int tm = r % TM, ks = r / TM;
    const int h = lane >> 5, lr = lane & 31;
    const int m = mb * BM + tm * 32 + lr;
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int o = kc * KC + ks * 16 + h * 8 + 2 * e2 + q;
      v[q] = (m < p.C3 && o < p.Cout) ? p.w[(long)o * p.C3 + m] : 0.f;
    }
    unsigned ph, pm, pl = 0;
    if constexpr (F16) split_pair_f16(v[0] * F16_W_SCALE, v[1] * F16_W_SCALE, ph, pm);
    else split_pair(v[0], v[1], ph, pm, pl);
    const int o2 = ((ks * TM + tm) * 64 + lane) * 8 + 2 * e2;
    *reinterpret_cast<unsigned*>(dst + 0 * PER_PLANE + o2) = ph;
    *reinterpret_cast<unsigned*>(dst + 1 * PER_PLANE + o2) = pm;
    if constexpr (!F16) *reinterpret_cast<unsigned*>(dst + 2 * PER_PLANE + o2) = pl;
  }
}

// F16: the projection H = Wd_i^T dy on f16x3 (split_f16.h): dy is multiplied by the power of two that brings max |dy| into
// [2^14, 2^15) while it is split, the partial sums by its inverse when they are stored.
template <int TM, int NW, bool F16 = false>
__global__ void __launch_bounds__(NW * 64, 2) gcn_dadj_chain_kernel(const DadjArgs a) {
  constexpr int NT = NW * 64, FT = NW, BM = TM * 32;
  constexpr int PL = F16 ? 2 : 3;
  constexpr int A_IMG = PL * 2 * TM * 1024;            // bytes of one stage's weight image
  constexpr int A16 = A_IMG / 16;
  constexpr int EA = (A16 + NT - 1) / NT;
  constexpr int BI = (FT * 32 * 4 + NT - 1) / NT;      // staging items (position, 8-channel group) per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int V = a.V, T = a.T;
  const int PT = FT * V;                               // positions of the frame tile
  const int B_IMG = PL * 4 * PT * 16;                  // bytes: [plane][ks][h][pos][8]
  float rs_s = 1.f, rs_inv = 1.f;
  // F16 with max |x| known: the reduction against x runs on f16x3 as well.  H = Wd_i^T dy is then split straight out of the
  // accumulators, so dy is scaled to 2^F16_ADJ_TARGET instead of 2^14: |H'| <= 2^(8 + TARGET + 1) * sum_o |w_oc| stays inside
  // fp16 for column sums of |Wd| below 32 (the reference's conv_d weights: ~1 at 64 channels, ~1-3 at 256; beyond: Inf -> NaN,
  // visible).  x is scaled to 2^14 like every directly split operand.
  const bool red16 = F16 && a.x_absmax != nullptr;     // kernel-uniform
  float xs = 1.f, xinv = 1.f;
  if constexpr (F16) {
    if (red16) {
      f16_range_scale_of<F16_ADJ_TARGET>(*a.dy_absmax, rs_s, rs_inv);
      f16_range_scale_of<14>(*a.x_absmax, xs, xinv);
    } else {
      f16_range_scale(a.dy_absmax, rs_s, rs_inv);
    }
    rs_inv *= F16_W_INV * xinv;
  }
  unsigned char* abuf = smem;                          // [2][A_IMG]
  unsigned char* bbuf = smem + 2 * A_IMG;              // [2][B_IMG]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mbk = bid % a.nmb;
  const int nt_id = bid / a.nmb;
  const int n = nt_id / a.ntiles, tile_id = nt_id - n * a.ntiles;
  const int m0 = mbk * BM;
  const int isub = mbk / a.gpc, c0 = (mbk - isub * a.gpc) * BM;   // subset and first channel of this row block
  const int t0 = tile_id * FT;
  const int t = t0 + wave;
  const bool fvalid = t < T;                           // wave-uniform
  const int plen = min(FT, T - t0) * V;                // valid positions of the tile
  const long P = (long)T * V;
  const int S = a.nkc;

  f32x16 acc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;

  // dy staging: lane task = (8-channel group og = ks*2 + h, 4 consecutive positions): eight 16-byte loads (one per
  // channel) instead of 32 scalar ones -- the per-CU address path, not HBM, bounded the scalar version.  Positions
  // beyond the tile's valid frames are never used (their waves skip), so nothing is masked; only the last positions of
  // the whole tensor take the clamped scalar path.
  u32x4 ra[EA];
  f32x4 rb[8];
  const u32x4* wp4 = reinterpret_cast<const u32x4*>(a.wp) + (long)mbk * S * A16;
  const int npiece = (PT + 3) >> 2, ntask = 4 * npiece;
  const float* dy_end = a.dy + (long)a.N * a.Cout * P;
  auto issue = [&](int s) __attribute__((always_inline)) {
    const u32x4* src = wp4 + (long)s * A16;
#pragma unroll
    for (int u = 0; u < EA; ++u) ra[u] = src[min(tid + u * NT, A16 - 1)];
    int tk = tid;
    asm volatile("" : "+v"(tk));                       // (keeps the task indices out of loop-invariant registers)
    if (tk < ntask) {
      const int og = tk / npiece, pos = (tk - og * npiece) * 4;
      const int o0 = s * KC + og * 8;
      const float* src2 = a.dy + ((long)n * a.Cout + min(o0, a.Cout - 8)) * P + (long)t0 * V + pos;
      if (src2 + 7 * P + 4 <= dy_end) {
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[e] = *reinterpret_cast<const f32x4_u*>(src2 + (long)e * P);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float* pq = src2 + (long)e * P + q;
            rb[e][q] = (pq < dy_end) ? *pq : 0.f;
          }
      }
    }
  };
  auto commit = [&](int s) __attribute__((always_inline)) {
    u32x4* dst = reinterpret_cast<u32x4*>(abuf + (s & 1) * A_IMG);
#pragma unroll
    for (int u = 0; u < EA; ++u)
      if (tid + u * NT < A16) dst[tid + u * NT] = ra[u];
    unsigned char* bd = bbuf + (s & 1) * B_IMG;
    int tk = tid;
    asm volatile("" : "+v"(tk));
    if (tk < ntask) {
      const int og = tk / npiece, pos = (tk - og * npiece) * 4;   // og = ks*2 + h
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x4 ph, pm, pl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          unsigned q0, q1, q2 = 0;
          if constexpr (F16) split_pair_f16(rb[2 * e2][q] * rs_s, rb[2 * e2 + 1][q] * rs_s, q0, q1);
          else split_pair(rb[2 * e2][q], rb[2 * e2 + 1][q], q0, q1, q2);
          ph[e2] = q0; pm[e2] = q1; pl[e2] = q2;
        }
        if (pos + q < PT) {
          *reinterpret_cast<u32x4*>(bd + ((0 * 4 + og) * PT + pos + q) * 16) = ph;
          *reinterpret_cast<u32x4*>(bd + ((1 * 4 + og) * PT + pos + q) * 16) = pm;
          if constexpr (!F16) *reinterpret_cast<u32x4*>(bd + ((2 * 4 + og) * PT + pos + q) * 16) = pl;
        }
      }
    }
  };

  issue(0);
  commit(0);
  if (S > 1) issue(1);
  __syncthreads();
  const int bpos = wave * V + min(lr, V - 1);          // this lane's column of the frame (padding lanes: clamped)
  // x operands of the reduction epilogue (raw loads, row u = lane, channel c_j + 4h of tile tm); tile 0 is fetched
  // under the last stage of the main loop so that its latency is hidden
  const float* xr = a.x + ((long)n * a.C + c0) * P + (long)min(t, T - 1) * V + min(lr, V - 1);
  auto load_x = [&](int tm, float (&xa)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = tm * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      xa[j] = xr[(long)min(c0 + c, a.C - 1) * P - (long)c0 * P];
    }
  };
  float xa0[16];
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) commit(s + 1);
    if (s + 2 < S) issue(s + 2);
    if (s == S - 1) load_x(0, xa0);
    if (fvalid && !(a.dbg & 2)) {
      const unsigned char* ab = abuf + (s & 1) * A_IMG + lane * 16;
      const unsigned char* bb = bbuf + (s & 1) * B_IMG;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 b[PL];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
          b[pl] = *reinterpret_cast<const bf16x8*>(bb + (((pl * 2 + ks) * 2 + h) * PT + bpos) * 16);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ab + ((0 * 2 + ks) * TM + tm) * 1024);
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ab + ((1 * 2 + ks) * TM + tm) * 1024);
          if constexpr (F16) {
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b[0]), acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b[1]), acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b[0]), acc[tm], 0, 0, 0);
            continue;
          }
          const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ab + ((PL - 1) * 2 + ks) * TM * 1024 + tm * 1024);
          if (a.npl == 3) {
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[0], acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[PL - 1], acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[1], acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[0], acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[1], acc[tm], 0, 0, 0);
          }
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[0], acc[tm], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- reduction against x: d[u][v] += sum_c x[c][t,u] * H[c][v]; the x operands come straight from global/L2
  // (row u = lane, k-pair of step j = channels c_j + 4h), one 16-load batch per 32-channel tile ----
  f32x16 d;
#pragma unroll
  for (int j = 0; j < 16; ++j) d[j] = 0.f;
  if (fvalid && !(a.dbg & 1)) {
    float xb[2][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (tm + 1 < TM) load_x(tm + 1, xb[(tm + 1) & 1]);          // next tile's operands in flight during this tile
      if (F16 && red16) {
        // k index e of step ks <-> D register 8 ks + e of this half: the accumulators ARE the B fragments once split, and
        // load_x fetched x in the same register order
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 ah, al, bh, bl;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const int j = 8 * ks + 2 * e2;
            float x0 = (tm == 0) ? xa0[j] : xb[tm & 1][j];
            float x1 = (tm == 0) ? xa0[j + 1] : xb[tm & 1][j + 1];
            x0 = (lr < V) ? x0 * xs : 0.f;
            x1 = (lr < V) ? x1 * xs : 0.f;
            unsigned p0, p1, q0, q1;
            split_pair_f16_mix(x0, x1, p0, p1);
            split_pair_f16_mix(acc[tm][j], acc[tm][j + 1], q0, q1);
            ah[e2] = p0; al[e2] = p1; bh[e2] = q0; bl[e2] = q1;
          }
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), d, 0, 0, 0);
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float xv = (tm == 0) ? xa0[j] : xb[tm & 1][j];
        d = mfma32((lr < V) ? xv : 0.f, acc[tm][j], d);
      }
    }
  }
  // ---- sum the NW frames of the tile and store the slot ----
  const int VV = V * V;
  float* red = reinterpret_cast<float*>(smem);         // [NW][VV]; the loop's final barrier freed LDS
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int u = mfma_row(j, h);
    if (u < V && lr < V) red[wave * VV + u * V + lr] = F16 ? d[j] * rs_inv : d[j];
  }
  __syncthreads();
  const int slot = tile_id * a.gpc + (mbk - isub * a.gpc);
  float* dst = a.dpart + (((long)n * 3 + isub) * a.nslots + slot) * VV;
  for (int e = tid; e < VV; e += NT) {
    float sum = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) sum += red[w2 * VV + e];
    dst[e] = sum;
  }
}

template <int TM, int NW, bool F16 = false>
int dadj_chain_launch(DadjArgs a, const float* wcat, void* ws, size_t ws_bytes, hipStream_t stream) {
  constexpr int BM = TM * 32, FT = NW;
  constexpr int PL = F16 ? 2 : 3;
  a.ntiles = (a.T + FT - 1) / FT;
  a.nkc = (a.Cout + KC - 1) / KC;
  a.gpc = a.C / BM;
  a.nmb = 3 * a.gpc;
  a.nslots = a.ntiles * a.gpc;
  const size_t a_img = (size_t)PL * 2 * TM * 1024;
  const size_t b_img = (size_t)PL * 4 * FT * a.V * 16;
  size_t smem_bytes = 2 * a_img + 2 * b_img;
  const size_t epi = (size_t)NW * a.V * a.V * 4;
  if (epi > smem_bytes) smem_bytes = epi;
  if (smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  const size_t pack_bytes = (size_t)a.nmb * a.nkc * a_img;
  if (pack_bytes + (F16 ? 16 : 0) > ws_bytes) return AGCN_ERR_WORKSPACE;
  a.wp = (const unsigned short*)ws;
  if constexpr (F16) {
    if (!a.dy_absmax) {      // behind the (two-plane) weight images of a workspace sized for three planes
      unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + pack_bytes);
      if (int rc = agcn_launch_absmax(a.dy, (long)a.N * a.Cout * a.T * a.V, amax, stream)) return rc;
      a.dy_absmax = reinterpret_cast<const float*>(amax);
    }
    static const int red16 = getenv("AGCN_DADJ_RED16") ? atoi(getenv("AGCN_DADJ_RED16")) : 1;   // 0: exact-f32 reduction (A/B)
    if (!red16) {
      a.x_absmax = nullptr;
    } else if (!a.x_absmax) {  // the same arithmetic with or without the producer's maximum: a pass of our own
      unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + pack_bytes) + 1;
      if (int rc = agcn_launch_absmax(a.x, (long)a.N * a.C * a.T * a.V, amax, stream)) return rc;
      a.x_absmax = reinterpret_cast<const float*>(amax);
    }
  }
  DadjPackArgs pk;
  pk.w = wcat; pk.wp = (unsigned short*)ws; pk.C3 = 3 * a.C; pk.Cout = a.Cout; pk.nkc = a.nkc;
  hipLaunchKernelGGL((dadj_pack_kernel<TM, F16>), dim3(a.nmb * a.nkc), dim3(256), 0, stream, pk);
  int rc = agcn_check_launch();
  if (rc) return rc;
  auto kern = gcn_dadj_chain_kernel<TM, NW, F16>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};   // per (kernel instantiation, device): the attribute is per device
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * a.ntiles * a.nmb)), dim3(NW * 64), smem_bytes, stream, a);
  return agcn_check_launch();
}

}  // namespace

// rows per block: 128 when M is a multiple of 128, 32 for the few-channel first layer, else 64
static inline int chain_tm(int M) { return (M % 128 == 0) ? 4 : (M <= 32 ? 1 : 2); }

// waves (= frames) per workgroup: 4 (two workgroups per CU overlap each other's prologue/epilogue) or 8
static inline int chain_waves() {
  static int nw = 0;
  if (!nw) {
    const char* e = getenv("AGCN_CHAIN_WAVES");
    nw = (e && atoi(e) == 8) ? 8 : 4;
  }
  return nw;
}

// 1 (default): aggregate+project forward / backward-data on f16x3 in the fp32-equivalent mode; AGCN_CHAIN_F16X3=0: bf16x6
int agcn_chain_f16x3() {
  static const int on = getenv("AGCN_CHAIN_F16X3") ? atoi(getenv("AGCN_CHAIN_F16X3")) : 1;
  return on && agcn_npl() == 3;
}

bool agcn_gcn_chain_supported(int M, int K, int V) { return M >= 1 && K >= 1 && V <= 32; }

int agcn_gcn_chain_tiles(int T) { return (T + chain_waves() - 1) / chain_waves(); }

// total (sum, sumsq) slots of the forward's BatchNorm partials: N * tiles for the tile-per-workgroup kernel, one per
// (sample, frame split) for the persistent one
int agcn_gcn_chain_stats_slots(int N, int M, int K, int T, int V) {
  if (ws_shape_ok(M, K, 0, V)) {
    const WsGeom g = (M <= 32) ? ws_geometry<1>(N, M, K, 0, T, V) : ws_geometry<2>(N, M, K, 0, T, V);
    if (g.smem_bytes <= 160 * 1024) return N * g.nsplit;
  }
  return N * agcn_gcn_chain_tiles(T);
}

// packed weight images; K2 = channels of the optional plain second source (0: none)
size_t agcn_gcn_chain_workspace(int M, int K, int K2, int T, int V) {
  (void)T; (void)V;
  const int tm = chain_tm(M), bm = 32 * tm;
  const size_t a_img = (size_t)3 * 2 * tm * 1024;
  return (size_t)((M + bm - 1) / bm) * (3 * ((K + CB - 1) / CB) + (K2 + CB - 1) / CB) * a_img;
}

// mode 0: forward (in = x, K = C, M = Cout); mode 1: backward-data (in = dy, K = Cout, M = C)
int agcn_gcn_chain(int mode, const float* in, const float* adj, const float* wcat, const float* bias, float* out,
                   float* stats_part, int accumulate, const float* add1, const float* mask1, const float* add2,
                   const float* mask2, int mask_bits, const float* in2, const float* w2, int K2, void* ws, size_t ws_bytes,
                   int N, int C, int Cout, int T, int V, hipStream_t stream, int relu, int w2_rows_are_outputs,
                   const float* in_absmax, const float* in2_absmax) {
  ChainArgs a = {};
  a.relu = relu;
  a.in_absmax = in_absmax; a.in2_absmax = in2_absmax;   // f16x3: maxima the producers left behind (null: a pre-pass)
  a.npl = agcn_npl();
  { static const int dbg = getenv("AGCN_GC_DBG") ? atoi(getenv("AGCN_GC_DBG")) : 0; a.dbg = dbg; }
  // optional fused 1x1 term (backward-data only): out += W2^T . in2 with w2 (K2, M) row-major, e.g. the theta/phi
  // branch  dx += Wab^T dtp  (reference agcn.py:99-100 differentiated)
  a.in2 = (in2 && w2 && K2 > 0) ? in2 : nullptr;
  a.K2 = a.in2 ? K2 : 0;
  ChainW2 cw2 = {w2, 1, 0};
  a.in = in; a.adj = adj; a.bias = bias; a.out = out; a.stats = stats_part;
  a.accumulate = accumulate; a.add1 = add1; a.mask1 = mask1; a.add2 = add2; a.mask2 = mask2; a.mask_bits = mask_bits;
  a.N = N; a.T = T; a.V = V;
  long sa_m, sa_i, sa_c;
  if (mode == 0) { a.M = Cout; a.K = C; a.adj_t = 0; sa_m = 3L * C; sa_i = C; sa_c = 1; }
  else           { a.M = C; a.K = Cout; a.adj_t = 1; sa_m = 1; sa_i = C; sa_c = 3L * C; }
  cw2.sa_c = a.M;        // W2[m][k] = w2[k*M + m]
  if (w2_rows_are_outputs) { cw2.sa_m = a.K2; cw2.sa_c = 1; }   // w2 (M, K2) row-major (forward: a folded 1x1 conv)
  if (!a.dbg && ws_shape_ok(a.M, a.K, a.K2, V)) {
    if (const char* e = getenv("AGCN_WS_DBG")) a.dbg = atoi(e);   // profiling switches of gcn_ws_kernel (read per call)              // persistent, weight-stationary kernel (64 streamed channels)
    const int rc = (a.M <= 32) ? ws_dispatch<1>(a, wcat, sa_m, sa_i, sa_c, cw2, ws, ws_bytes, stream)
                               : ws_dispatch<2>(a, wcat, sa_m, sa_i, sa_c, cw2, ws, ws_bytes, stream);
    if (rc != AGCN_ERR_UNSUPPORTED) return rc;
  }
  if (chain_waves() == 8 && chain_tm(a.M) != 1) {
    if (chain_tm(a.M) == 4) return chain_dispatch_vs<4, 8>(a, wcat, sa_m, sa_i, sa_c, cw2, ws, ws_bytes, stream);
    return chain_dispatch_vs<2, 8>(a, wcat, sa_m, sa_i, sa_c, cw2, ws, ws_bytes, stream);
  }
  if (chain_tm(a.M) == 4) return chain_dispatch_vs<4, 4>(a, wcat, sa_m, sa_i, sa_c, cw2, ws, ws_bytes, stream);
  if (chain_tm(a.M) == 1) return chain_dispatch_vs<1, 4>(a, wcat, sa_m, sa_i, sa_c, cw2, ws, ws_bytes, stream);
  return chain_dispatch_vs<2, 4>(a, wcat, sa_m, sa_i, sa_c, cw2, ws, ws_bytes, stream);
}

// ---- adjacency gradient (gcn_dadj_chain_kernel): C a multiple of 64; row block 128 when C is a multiple of 128 ----
constexpr int DADJ_NW = 8;    // 128-row blocks: 8 frames per workgroup (one per CU)
constexpr int DADJ_NW64 = 4;  // 64-row blocks: 4 frames, two workgroups per CU
bool agcn_gcn_dadj_chain_supported(int C, int V) { return C >= 64 && C % 64 == 0 && V <= 32; }

int agcn_gcn_dadj_chain_slots(int C, int T) {
  const int bm = (C % 128 == 0) ? 128 : 64;
  const int nw = (bm == 128) ? DADJ_NW : DADJ_NW64;
  return ((T + nw - 1) / nw) * (C / bm);
}

size_t agcn_gcn_dadj_chain_workspace(int C, int Cout) {
  const int bm = (C % 128 == 0) ? 128 : 64;
  return (size_t)(3 * C / bm) * ((Cout + KC - 1) / KC) * 3 * 2 * (bm / 32) * 1024;
}

int agcn_gcn_dadj_chain(const float* dy, const float* wcat, const float* x, float* dadj_part, void* ws, size_t ws_bytes,
                        int N, int C, int Cout, int T, int V, hipStream_t stream, const float* dy_absmax,
                        const float* x_absmax) {
  DadjArgs a = {};
  a.npl = agcn_npl();
  a.dy = dy; a.x = x; a.dpart = dadj_part; a.N = N; a.C = C; a.Cout = Cout; a.T = T; a.V = V;
  a.dy_absmax = dy_absmax;
  a.x_absmax = x_absmax;       // (null: taken by a pass inside; AGCN_DADJ_RED16=0: exact-f32 reduction MFMA)
  if (const char* e = getenv("AGCN_DADJ_DBG")) a.dbg = atoi(e);
  static const int f16 = getenv("AGCN_DADJ_F16X3") ? atoi(getenv("AGCN_DADJ_F16X3")) : 1;   // 0: bf16x6 (A/B)
  if (agcn_chain_f16x3() && f16) {
    if (C % 128 == 0) return dadj_chain_launch<4, DADJ_NW, true>(a, wcat, ws, ws_bytes, stream);
    return dadj_chain_launch<2, DADJ_NW64, true>(a, wcat, ws, ws_bytes, stream);
  }
  if (C % 128 == 0) return dadj_chain_launch<4, DADJ_NW>(a, wcat, ws, ws_bytes, stream);
  return dadj_chain_launch<2, DADJ_NW64>(a, wcat, ws, ws_bytes, stream);
}
