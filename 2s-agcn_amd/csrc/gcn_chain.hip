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
    if constexpr (F16) split_pair_f16(v[0], v[1], ph, pm);
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
// products; the staged source is multiplied by the power of two that brings max(|in|, |in2|) into [2^8, 2^9) before it
// is split (so that G = x . A^ stays inside fp16's range for column sums of |A^| up to 2^7) and the accumulators by its
// inverse in the epilogue.
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
  float* xb = reinterpret_cast<float*>(smem + ADJ_BYTES);             // [CB][XP] (+ 8 floats of slack when BCH)
  const int xb_bytes = ((CB * a.XP * 4 + (BCH ? 32 : 0) + 15) & ~15);
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
    f16_range_scale_of<8>(mx, rs_s, rs_inv);
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
    if (tid < 8) xb[CB * XP + tid] = 0.f;                            // slack behind the last chunk row
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
  const size_t xb_bytes = ((size_t)CB * g.XP * 4 + (bch ? 32 : 0) + 15) & ~(size_t)15;
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
    if constexpr (F16) split_pair_f16(v[0], v[1], ph, pm);
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
  if constexpr (F16) f16_range_scale(a.dy_absmax, rs_s, rs_inv);
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
    if (fvalid) {
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
  if (fvalid) {
    float xb[2][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (tm + 1 < TM) load_x(tm + 1, xb[(tm + 1) & 1]);          // next tile's operands in flight during this tile
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
                        int N, int C, int Cout, int T, int V, hipStream_t stream, const float* dy_absmax) {
  DadjArgs a = {};
  a.npl = agcn_npl();
  a.dy = dy; a.x = x; a.dpart = dadj_part; a.N = N; a.C = C; a.Cout = Cout; a.T = T; a.V = V;
  a.dy_absmax = dy_absmax;
  static const int f16 = getenv("AGCN_DADJ_F16X3") ? atoi(getenv("AGCN_DADJ_F16X3")) : 1;   // 0: bf16x6 (A/B)
  if (agcn_chain_f16x3() && f16) {
    if (C % 128 == 0) return dadj_chain_launch<4, DADJ_NW, true>(a, wcat, ws, ws_bytes, stream);
    return dadj_chain_launch<2, DADJ_NW64, true>(a, wcat, ws, ws_bytes, stream);
  }
  if (C % 128 == 0) return dadj_chain_launch<4, DADJ_NW>(a, wcat, ws, ws_bytes, stream);
  return dadj_chain_launch<2, DADJ_NW64>(a, wcat, ws, ws_bytes, stream);
}
