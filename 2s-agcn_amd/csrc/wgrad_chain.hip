// Weight gradients of the tap-free channel contractions on split-bf16 MFMA (bf16x6, fp32-equivalent):
//   AGG = 1:  dWd_i[o][c] = sum_{n,t,v} dy[n][o][t,v] * (x . A^_i)[n][c][t,v]      (unit_gcn projection, agcn.py:103-105)
//   AGG = 0:  dW[o][c]    = sum_{n,t,v} dy[n][o][t,v] * x[n][c][t*stride, v]       (1x1 convs: conv_a/b, down, residual)
// GEMM view: M = output channels o, N = input channels c, K = positions (n, t, v).
//
// Mapping (same register chaining as gcn_chain.hip): a wave owns one 32-channel block of c and a subset of the
// frames of the staged tile.  Per frame it builds the TRANSPOSED operand G^T[v][c] in MFMA D layout -- by an exact-f32
// MFMA chain A^_i^T . x^T (AGG) or by plain LDS reads (AGG = 0) -- so register j of lane (h, c) holds joint
// v = (j&3) + 8*(j>>2) + 4*h of channel c: split in registers into bf16 (hi, mid, lo) these 16 values are the B
// operand (K = 32 padded joints = two 16-deep steps) of the MFMA against dy, whose LDS image the staging code wrote
// already split and in the same permuted joint order.  The aggregated operand never exists outside registers.
// Waves = NCB channel blocks x NFG frame groups; every frame group writes its own partial slab (split-K), summed by
// the fixed-order wgrad_reduce kernel of conv_wgrad.hip.
#include "agcn_common.h"
#include "split_f16.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));     // rows of V floats are only 4-byte aligned

struct WcArgs {
  const float* dy;     // (N, M, T_out, V)
  const float* in;     // (N, C, T_src, V)
  const float* adj;    // (N, 3, V, V) or null
  float* part;         // [slab][z][M][C]
  int N, M, C, V, T_src, T_out, stride;
  int ntiles, pairs_per_split, ncg;   // frame tiles per sample, (sample, tile) pairs per blockIdx.y, channel groups
  int XP;                              // pitch (floats) of a staged x row (odd)
  long wsize;
  int npl;             // 3: six split products (bf16x6) ; 1: hi*hi only (bf16)
  int dbg;             // profiling switches (AGCN_WC_DBG): 1 = no matrix work, 2 = stage the first pair only, 4 = no f32 chain
  int pad;             // 1: [o] blocks of the dy image padded by one slot (AGCN_WC_PAD=0 for the A/B measurement)
  const float* dy_absmax;   // f16x3 (wgrad_pc_kernel<..., true>): device scalars max |dy|, max |in| for the range scales
  const float* in_absmax;
};

__device__ __forceinline__ unsigned wc_pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 p = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void wc_split_pair(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
  ph = wc_pack_bf16(a, b);
  const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
  pm = wc_pack_bf16(ra, rb);
  pl = wc_pack_bf16(ra - __builtin_bit_cast(float, pm << 16), rb - __builtin_bit_cast(float, pm & 0xffff0000u));
}
// joint carried by slot e of lane-half h in 16-deep step ks (D register j = 8*ks + e)
__device__ __forceinline__ int wc_joint(int ks, int h, int e) {
  const int j = 8 * ks + e;
  return (j & 3) + 8 * (j >> 2) + 4 * h;
}

// TM: 32-row tiles of o per workgroup (BM = 32*TM); NCB: channel blocks (waves) per workgroup; VS: aggregation steps
// NW = waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU, which overlap each other's staging and barriers)
template <int AGG, int TM, int NCB, int VS, int NW = 8>
__global__ void __launch_bounds__(NW * 64, 2) wgrad_chain_kernel(const WcArgs a) {
  constexpr int NT = NW * 64, BM = TM * 32;
  constexpr int NFG = NW / NCB;                 // frame groups
  constexpr int FPW = (NCB <= 2) ? 1 : 2;       // frames per wave per stage
  constexpr int FT = NFG * FPW;                 // frames per stage
  constexpr int CG = NCB * 32;                  // channels per workgroup
  constexpr int DI = BM * FT * 4 / NT;          // dy staging items (o, f, ks, h) per thread
  static_assert(BM * FT * 4 % NT == 0, "dy items must tile the threads");
  constexpr int XR = CG / NW;                   // x rows per wave
  constexpr int XB = (FT * 32 + 63) / 64;       // 64-float column blocks of an x row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [plane][f][ks][h][o (BMP slots)][8] bf16.  Each [o] block is padded by one 16-byte slot: the staging threads walk
  // (f, ks, h) fastest, i.e. from block to block, and with a power-of-two block every ds_write_b128 of a wave hit the
  // same banks (measured: 60 % of the kernel's LDS cycles were bank conflicts); BM+1 slots put the 8 lanes of a
  // write group on 8 different 16-byte columns.  Fragment reads walk o (consecutive slots) and are unaffected.
  const int BMP = BM + a.pad;
  unsigned char* dyi = smem;
  const int DY_BYTES = 3 * FT * 4 * BMP * 16;
  float* xs = reinterpret_cast<float*>(smem + DY_BYTES);            // [CG][XP]
  float* adjp = xs + CG * a.XP;                                     // [32][32] (AGG)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int cbw = wave % NCB, fg = wave / NCB;
  // blockIdx.x -> (subset, row block, channel group)
  const int isub = AGG ? (int)(blockIdx.x % 3) : 0;
  const int rest = AGG ? (int)(blockIdx.x / 3) : (int)blockIdx.x;
  const int cgp = rest % a.ncg, mb = rest / a.ncg;
  const int m0 = mb * BM, c0 = cgp * CG;
  const int V = a.V, XP = a.XP;
  const long Pout = (long)a.T_out * V, Psrc = (long)a.T_src * V;

  f32x16 acc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;

  float rdy[DI][8];
  float rx[XR][XB];
  auto pair_geom = [&](int p, int& n, int& t0) __attribute__((always_inline)) {
    n = p / a.ntiles;
    t0 = (p - n * a.ntiles) * FT;
  };
  // (tid is laundered inside the staging code: otherwise the per-thread item indices are hoisted out of the pair loop
  // and held in registers across the matrix phase, which spills)
  auto issue = [&](int p) __attribute__((always_inline)) {
    int n, t0;
    pair_geom(p, n, t0);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
#pragma unroll
    for (int k = 0; k < DI; ++k) {
      const int item = tid + k * NT;                 // = o*(FT*4) + f*4 + (ks*2 + hh)
      const int o = item / (FT * 4), r = item - o * (FT * 4);
      const int f = r >> 2, ks = (r >> 1) & 1, hh = r & 1;
      const bool ok = (m0 + o) < a.M && (t0 + f) < a.T_out;
      const float* src = a.dy + ((long)n * a.M + (ok ? (m0 + o) : 0)) * Pout + (long)(ok ? (t0 + f) : 0) * V;
      // the 8 joints of an item are two runs of 4 consecutive joints (wc_joint): one 16-byte load each when the run
      // lies inside the frame, clamped scalar loads for a run that crosses the last joint
#pragma unroll
      for (int run = 0; run < 2; ++run) {
        const int v0 = 16 * ks + 4 * hh + 8 * run;
        if (v0 + 3 < V) {
          const f32x4 q = *reinterpret_cast<const f32x4_u*>(src + v0);
#pragma unroll
          for (int e = 0; e < 4; ++e) rdy[k][4 * run + e] = q[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rdy[k][4 * run + e] = src[min(v0 + e, V - 1)];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int c = c0 + wave * XR + j;
      const float* src = a.in + ((long)n * a.C + min(c, a.C - 1)) * Psrc;
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        const int f = q / V, v = q - f * V;
        const bool ok = q < FT * V && (t0 + f) < a.T_out;
        rx[j][u] = src[ok ? ((long)(t0 + f) * a.stride * V + v) : 0];
      }
    }
  };
  auto commit = [&](int p) __attribute__((always_inline)) {
    int n, t0;
    pair_geom(p, n, t0);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
#pragma unroll
    for (int k = 0; k < DI; ++k) {
      const int item = tid + k * NT;
      const int o = item / (FT * 4), r = item - o * (FT * 4);
      const int f = r >> 2, ks = (r >> 1) & 1, hh = r & 1;
      const bool ok = (m0 + o) < a.M && (t0 + f) < a.T_out;
      u32x4 ph, pm, pl;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const float v0 = (ok && wc_joint(ks, hh, 2 * e2) < V) ? rdy[k][2 * e2] : 0.f;
        const float v1 = (ok && wc_joint(ks, hh, 2 * e2 + 1) < V) ? rdy[k][2 * e2 + 1] : 0.f;
        unsigned q0, q1, q2;
        wc_split_pair(v0, v1, q0, q1, q2);
        ph[e2] = q0; pm[e2] = q1; pl[e2] = q2;
      }
      const int slot = ((f * 2 + ks) * 2 + hh) * BMP + o;
      *reinterpret_cast<u32x4*>(dyi + ((0 * FT * 4) * BMP + slot) * 16) = ph;
      *reinterpret_cast<u32x4*>(dyi + ((1 * FT * 4) * BMP + slot) * 16) = pm;
      *reinterpret_cast<u32x4*>(dyi + ((2 * FT * 4) * BMP + slot) * 16) = pl;
    }
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int cl = wave * XR + j;
      const bool rok = (c0 + cl) < a.C;
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = lane + 64 * u;
        const int f = q / V;
        if (q < FT * V) xs[cl * XP + q] = (rok && (t0 + f) < a.T_out) ? rx[j][u] : 0.f;
      }
    }
  };

  const int total_pairs = a.N * a.ntiles;
  const int p_begin = blockIdx.y * a.pairs_per_split;
  const int p_end = min(total_pairs, p_begin + a.pairs_per_split);
  int last_n = -1;
  if (p_begin < p_end) issue(p_begin);
  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();                          // every wave is done with the previous pair's LDS tiles
    const bool stage = !(a.dbg & 2) || p == p_begin;
    if (stage) commit(p);
    int n, t0;
    pair_geom(p, n, t0);
    if (AGG && n != last_n && stage) {
      // zero-padded adjacency of this subset, adjp[u][v]
      const float* adjn = a.adj + ((long)n * 3 + isub) * V * V;
      for (int e = tid; e < 32 * 32; e += NT) {
        const int u = e >> 5, v = e & 31;
        const bool ok = u < V && v < V;
        const float tv = adjn[ok ? (u * V + v) : 0];
        adjp[e] = ok ? tv : 0.f;
      }
      last_n = n;
    }
    if (p + 1 < p_end && stage) issue(p + 1); // in flight during this pair's matrix-core work
    __syncthreads();
    if (a.dbg & 1) continue;
#pragma unroll 1
    for (int k = 0; k < FPW; ++k) {
      const int f = fg + k * NFG;
      if (t0 + f >= a.T_out) continue;        // wave-uniform
      // ---- G^T[v][c] for (frame f, this wave's 32 channels) in D layout ----
      f32x16 d;
      const float* xr = xs + (cbw * 32 + lr) * XP + f * V;
      if (AGG && !(a.dbg & 4)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = 0.f;
        float ao[VS], xo[VS];
#pragma unroll
        for (int s = 0; s < VS; ++s) {
          ao[s] = adjp[(2 * s + h) * 32 + lr];                 // A operand: A^[u = 2s+h][v = lane]
          xo[s] = xr[min(2 * s + h, V - 1)];                   // B operand: x[c = lane][t][u = 2s+h]
        }
#pragma unroll
        for (int s = 0; s < VS; ++s) d = mfma32(ao[s], xo[s], d);
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int v = (j & 3) + 8 * (j >> 2) + 4 * h;
          const float t = xr[min(v, V - 1)];
          d[j] = (v < V) ? t : 0.f;
        }
      }
      // ---- split and contract with dy: acc[tm][o][c] += sum_v dy[o][v] * G^T[v][c] ----
      bf16x8 gb[2][3];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 gh, gm, gl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          unsigned q0, q1, q2;
          wc_split_pair(d[8 * ks + 2 * e2], d[8 * ks + 2 * e2 + 1], q0, q1, q2);
          gh[e2] = q0; gm[e2] = q1; gl[e2] = q2;
        }
        gb[ks][0] = __builtin_bit_cast(bf16x8, gh);
        gb[ks][1] = __builtin_bit_cast(bf16x8, gm);
        gb[ks][2] = __builtin_bit_cast(bf16x8, gl);
      }
      // 2*TM steps of 6 MFMAs; the dy fragments of step s+1 are read while step s runs (order pinned below)
      const unsigned char* ab = dyi + (((f * 4 + h) * BMP) + lr) * 16;
      auto load_a = [&](bf16x8 (&af)[3], int step) __attribute__((always_inline)) {
        const int ks = step / TM, tm = step - ks * TM;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          af[pl] = *reinterpret_cast<const bf16x8*>(ab + ((pl * FT * 4 + ks * 2) * BMP + tm * 32) * 16);
      };
      bf16x8 afc[3];
      load_a(afc, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int step = 0; step < 2 * TM; ++step) {
        const int ks = step / TM, tm = step - ks * TM;
        bf16x8 afn[3];
        if (step + 1 < 2 * TM) load_a(afn, step + 1);
        if (a.npl == 3) {
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[2], gb[ks][0], acc[tm], 0, 0, 0);
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[0], gb[ks][2], acc[tm], 0, 0, 0);
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[1], gb[ks][1], acc[tm], 0, 0, 0);
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[1], gb[ks][0], acc[tm], 0, 0, 0);
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[0], gb[ks][1], acc[tm], 0, 0, 0);
        }
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[0], gb[ks][0], acc[tm], 0, 0, 0);
        if (step + 1 < 2 * TM) {
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) afc[pl] = afn[pl];
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
    }
  }
  // ---- this frame group's partial slab, [z][m][c] (lanes = consecutive c: coalesced rows) ----
  const int slab = (int)blockIdx.y * NFG + fg;
  float* dst = a.part + (long)slab * a.wsize + (long)isub * a.M * a.C;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int m = m0 + tm * 32 + mfma_row(j, h);
      const int c = c0 + cbw * 32 + lr;
      if (m < a.M && c < a.C) dst[(long)m * a.C + c] = acc[tm][j];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Producer / consumer variant (C a multiple of 64).  Measured on the kernel above: staging (global loads, the bf16
// split of dy, LDS writes: ~700 vector instructions per thread and stage) and the matrix phase were ADDITIVE, because
// every wave did both between the same two barriers and the matrix pipe sat idle while all of them staged.  Here the
// workgroup has NWC consumer waves, which only run the chain + contraction of stage p out of LDS buffer p&1, and NWP
// producer waves, which meanwhile split stage p+1 into the other buffer (their global loads are issued one more stage
// ahead, into registers).  One barrier per stage; the producers' vector work now overlaps the consumers' MFMAs on the
// same SIMDs (3 waves per SIMD: 2 consumers + 1 producer).
// AGG: the workgroup handles all three subsets (dy and x staged once, a third of the HBM/L2 traffic of one workgroup per
// subset: measured, that variant ran at the memory system's pace); a consumer wave then holds 3*TM accumulator tiles.
// NRB: 32*TM-row blocks per workgroup (consumer waves = NCB channel blocks x NFG frames x NRB row blocks)
// F16: the contraction on f16x3 (split_f16.h): dy is scaled so that max |dy| lands in [2^14, 2^15) while the producers
// split it, G (= x, or x . A^ from the exact-f32 chain) so that max |x| lands in [2^8, 2^9) (headroom for the adjacency's
// column sums) while the consumers split it; the slabs are un-scaled when stored.
template <int AGG, int TM, int NCB, int VS, int NRB = 1, int NWC = 8, int NWP = 4, bool F16 = false>
__global__ void __launch_bounds__((NWC + NWP) * 64, (NWC + NWP) / 4) wgrad_pc_kernel(const WcArgs a) {
  constexpr int PL = F16 ? 2 : 3, NPROD = F16 ? 3 : 6;
  constexpr int NTP = NWP * 64, BM = TM * 32 * NRB;
  constexpr int NS = AGG ? 3 : 1;
  constexpr int NFG = NWC / (NCB * NRB);        // frame groups = frames per stage (one frame per consumer wave)
  constexpr int FT = NFG;
  constexpr int CG = NCB * 32;
  constexpr int DI = BM * FT * 4 / NTP;         // dy items (o, f, ks, h) per producer thread
  static_assert(BM * FT * 4 % NTP == 0, "dy items must tile the producer threads");
  constexpr int BMP = BM + 1;
  constexpr int DY_BYTES = PL * FT * 4 * BMP * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int V = a.V, XP = a.XP;
  const int FTV = FT * V;
  float s_dy = 1.f, inv_dy = 1.f, s_x = 1.f, inv_x = 1.f;
  if constexpr (F16) {
    f16_range_scale_of<14>(*a.dy_absmax, s_dy, inv_dy);
    f16_range_scale_of<F16_ADJ_TARGET>(*a.in_absmax, s_x, inv_x);
  }
  // VS == 0: the aggregation chain runs on split-bf16 MFMA (12 MFMAs of 32 cycles for K = 32 joints instead of VS
  // exact-f32 steps of 64 cycles): adjacencies kept as bf16 planes [subset][plane][ks][h][v][8 u] (gcn_chain.hip)
  // F16 with VS == 0: the aggregation on f16x3 as well (6 MFMAs of 32 cycles instead of 13 x 64: the exact-f32 chain was
  // two thirds of the consumers' matrix-pipe time): x is scaled to 2^F16_ADJ_TARGET BEFORE the aggregation, G comes out
  // scaled and is split as it is; adjacencies as two fp16 planes
  constexpr bool BCH = AGG && VS == 0;
  constexpr int PLA = F16 ? 2 : 3;                   // adjacency planes
  constexpr int ADJ_BYTES = !AGG ? 0 : (BCH ? 3 * PLA * 2 * 2 * 32 * 16 : 3 * 32 * 32 * 4);
  // BCH: the last frame's 32-joint fragment of the last x row runs up to 32 - V floats past the row: slack, zeroed once
  // (what it meets is a zero row of the adjacency, but stale bits must not turn into Inf / NaN when converted to fp16)
  constexpr int XSLACK = BCH ? 128 : 0;
  const int BUF_BYTES = DY_BYTES + ((CG * XP * 4 + 15) & ~15) + XSLACK + ADJ_BYTES;
  const long Pout = (long)a.T_out * V, Psrc = (long)a.T_src * V;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rest = (int)blockIdx.x;
  const int cgp = rest % a.ncg, mb = rest / a.ncg;
  const int m0 = mb * BM, c0 = cgp * CG;
  const int total_pairs = a.N * a.ntiles;
  const int p_begin = blockIdx.y * a.pairs_per_split;
  const int p_end = min(total_pairs, p_begin + a.pairs_per_split);
  auto pair_geom = [&](int p, int& n, int& t0) __attribute__((always_inline)) {
    n = p / a.ntiles;
    t0 = (p - n * a.ntiles) * FT;
  };
  // every wave of the workgroup executes exactly 1 + (p_end - p_begin) of these
  auto stage_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  if (wave >= NWC) {
    // =============================================== producers ===============================================
    constexpr int LPR = FT <= 2 ? 16 : 32;        // lanes per x row: (FT*V + 3) / 4 pieces of 16 bytes, V <= 32
    constexpr int XV = CG / NWP / (64 / LPR);     // 16-byte x loads per producer thread
    static_assert(CG % (NWP * (64 / LPR)) == 0, "x rows must tile the producer waves");
    float rdy[DI][8];
    f32x4 rx[XV];
    int adj_n[2] = {-1, -1};
    // Loads are 16 bytes wide and NOT masked: what a lane reads beyond the joints / frames it needs is a neighbouring
    // frame or row of the same tensor (finite), and it only ever meets zeros (rows of G^T beyond V are exactly 0; frames
    // beyond T_out and rows beyond M / C are skipped or not stored).  Only the last frame(s) of the whole tensor
    // would read past its end: those lanes take the clamped scalar path (one wave of the grid at most).
    const float* dy_end = a.dy + (long)a.N * a.M * Pout;
    const float* x_end = a.in + (long)a.N * a.C * Psrc;
    auto issue = [&](int p) __attribute__((always_inline)) {
      int n, t0;
      pair_geom(p, n, t0);
      int ptid = threadIdx.x - NWC * 64;
      asm volatile("" : "+v"(ptid));             // (see the note on laundering above)
#pragma unroll
      for (int k = 0; k < DI; ++k) {
        const int item = ptid + k * NTP;         // = o*(FT*4) + f*4 + (ks*2 + hh)
        const int o = item / (FT * 4), r = item - o * (FT * 4);
        const int f = r >> 2, ks = (r >> 1) & 1, hh = r & 1;
        const bool ok = (m0 + o) < a.M && (t0 + f) < a.T_out;
        const float* src = a.dy + ((long)n * a.M + (ok ? (m0 + o) : 0)) * Pout + (long)(ok ? (t0 + f) : 0) * V +
                           16 * ks + 4 * hh;
        if (src + 12 <= dy_end) {
#pragma unroll
          for (int run = 0; run < 2; ++run) {
            const f32x4 q = *reinterpret_cast<const f32x4_u*>(src + 8 * run);
#pragma unroll
            for (int e = 0; e < 4; ++e) rdy[k][4 * run + e] = q[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float* q = src + (e & 3) + 8 * (e >> 2);
            rdy[k][e] = (q < dy_end) ? *q : 0.f;
          }
        }
      }
      const int piece = ptid & (LPR - 1), rsub = (ptid & 63) / LPR;
      const float* xb = a.in + (long)n * a.C * Psrc + (long)t0 * V + 4 * piece;
#pragma unroll
      for (int u = 0; u < XV; ++u) {
        const int row = (wave - NWC) * (CG / NWP) + u * (64 / LPR) + rsub;
        const float* src = xb + (long)min(c0 + row, a.C - 1) * Psrc;
        if (src + 4 <= x_end) {
          rx[u] = *reinterpret_cast<const f32x4_u*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rx[u][e] = (src + e < x_end) ? src[e] : 0.f;
        }
      }
    };
    auto commit = [&](int p, unsigned char* buf) __attribute__((always_inline)) {
      int n, t0;
      pair_geom(p, n, t0);
      int ptid = threadIdx.x - NWC * 64;
      asm volatile("" : "+v"(ptid));
      unsigned char* dyi = buf;
      float* xs = reinterpret_cast<float*>(buf + DY_BYTES);
#pragma unroll
      for (int k = 0; k < DI; ++k) {
        const int item = ptid + k * NTP;
        const int o = item / (FT * 4), r = item - o * (FT * 4);
        u32x4 ph, pm, pl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          unsigned q0, q1, q2 = 0;
          if constexpr (F16) split_pair_f16(rdy[k][2 * e2] * s_dy, rdy[k][2 * e2 + 1] * s_dy, q0, q1);
          else wc_split_pair(rdy[k][2 * e2], rdy[k][2 * e2 + 1], q0, q1, q2);
          ph[e2] = q0; pm[e2] = q1; pl[e2] = q2;
        }
        const int slot = r * BMP + o;             // r = (f*2 + ks)*2 + hh
        *reinterpret_cast<u32x4*>(dyi + ((0 * FT * 4) * BMP + slot) * 16) = ph;
        *reinterpret_cast<u32x4*>(dyi + ((1 * FT * 4) * BMP + slot) * 16) = pm;
        if constexpr (!F16) *reinterpret_cast<u32x4*>(dyi + ((2 * FT * 4) * BMP + slot) * 16) = pl;
      }
      const int piece = ptid & (LPR - 1), rsub = (ptid & 63) / LPR;
#pragma unroll
      for (int u = 0; u < XV; ++u) {
        const int row = (wave - NWC) * (CG / NWP) + u * (64 / LPR) + rsub;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = 4 * piece + e;
          if (q < FTV) xs[row * XP + q] = rx[u][e];
        }
      }
      if (AGG && n != adj_n[p & 1]) {
        // zero-padded adjacencies of this sample, adjp[subset][u][v] (a buffer keeps them while the sample lasts)
        adj_n[p & 1] = n;
        const float* adjn = a.adj + (long)n * 3 * V * V;
        if constexpr (!BCH) {
          float* adjp = reinterpret_cast<float*>(buf + BUF_BYTES - ADJ_BYTES);
          for (int e = ptid; e < 3 * 32 * 32; e += NTP) {
            const int sub = e >> 10, u = (e >> 5) & 31, v = e & 31;
            const bool ok = u < V && v < V;
            const float tv = adjn[ok ? ((sub * V + u) * V + v) : 0];
            adjp[e] = ok ? tv : 0.f;
          }
        } else {
          unsigned char* adjq = buf + BUF_BYTES - ADJ_BYTES;
          for (int e = ptid; e < 3 * 2 * 2 * 32 * 4; e += NTP) {       // one pair (u, u+1) per iteration
            const int e2 = e & 3, col = (e >> 2) & 31, hh = (e >> 7) & 1, ks = (e >> 8) & 1, sub = e >> 9;
            float vv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int u = 16 * ks + 8 * hh + 2 * e2 + q;
              const bool ok = u < V && col < V;
              const float tv = adjn[ok ? ((sub * V + u) * V + col) : 0];
              vv[q] = ok ? tv : 0.f;
            }
            unsigned q0, q1, q2 = 0;
            if constexpr (F16) split_pair_f16(vv[0], vv[1], q0, q1);
            else wc_split_pair(vv[0], vv[1], q0, q1, q2);
            const int o = (((ks * 2 + hh) * 32) + col) * 16 + e2 * 4;
            *reinterpret_cast<unsigned*>(adjq + (sub * PLA + 0) * 2048 + o) = q0;
            *reinterpret_cast<unsigned*>(adjq + (sub * PLA + 1) * 2048 + o) = q1;
            if constexpr (!F16) *reinterpret_cast<unsigned*>(adjq + (sub * PLA + 2) * 2048 + o) = q2;
          }
        }
      }
    };
    if (XP > FTV)                                // the pad element of a row is read (times zero) by the chain
      for (int e = threadIdx.x - NWC * 64; e < 2 * CG; e += NTP)
        reinterpret_cast<float*>(smem + (e / CG) * BUF_BYTES + DY_BYTES)[(e % CG) * XP + FTV] = 0.f;
    {                                            // the alignment gap (+ slack) behind the last x row is read too (times zero)
      const int gap = (((CG * XP * 4 + 15) & ~15) - CG * XP * 4 + XSLACK) / 4;
      const int e = threadIdx.x - NWC * 64;
      if (e < 2 * gap) reinterpret_cast<float*>(smem + (e / max(gap, 1)) * BUF_BYTES + DY_BYTES)[CG * XP + e % max(gap, 1)] = 0.f;
    }
    if (p_begin < p_end) {
      issue(p_begin);
      commit(p_begin, smem + (p_begin & 1) * BUF_BYTES);
      if (p_begin + 1 < p_end) issue(p_begin + 1);
    }
    stage_barrier();
    for (int p = p_begin; p < p_end; ++p) {
      if (p + 1 < p_end && !(a.dbg & 2)) {
        commit(p + 1, smem + ((p + 1) & 1) * BUF_BYTES);
        if (p + 2 < p_end) issue(p + 2);
      }
      stage_barrier();
    }
    return;
  }

  // ================================================= consumers =================================================
  const int lane = threadIdx.x & 63;
  const int lr = lane & 31, h = lane >> 5;
  const int cbw = wave % NCB, fg = (wave / NCB) % NFG, rb = wave / (NCB * NFG);
  f32x16 acc[NS][TM];
#pragma unroll
  for (int sub = 0; sub < NS; ++sub)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[sub][tm][j] = 0.f;
  stage_barrier();
  for (int p = p_begin; p < p_end; ++p) {
    int n, t0;
    pair_geom(p, n, t0);
    const unsigned char* dyi = smem + (p & 1) * BUF_BYTES;
    const float* xs = reinterpret_cast<const float*>(dyi + DY_BYTES);
    const int f = fg;
    if (t0 + f < a.T_out && !(a.dbg & 1)) {    // wave-uniform
      const float* xr = xs + (cbw * 32 + lr) * XP + f * V;
      const unsigned char* ab = dyi + (((f * 4 + h) * BMP) + rb * (TM * 32) + lr) * 16;
#pragma unroll
      for (int sub = 0; sub < NS; ++sub) {
        // ---- G^T[v][c] for (frame f, this wave's 32 channels, subset sub) in D layout ----
        f32x16 d;
        if (AGG) {
#pragma unroll
          for (int j = 0; j < 16; ++j) d[j] = 0.f;
          // (x is re-read for every subset: kept across the subsets its registers push the contraction into spills)
          const float* xrs = xr;
          asm volatile("" : "+v"(xrs));
          if constexpr (!BCH) {
            const float* adjp = reinterpret_cast<const float*>(dyi + BUF_BYTES - ADJ_BYTES) + sub * 1024;
            float ao[BCH ? 1 : VS], xo[BCH ? 1 : VS];
#pragma unroll
            for (int s = 0; s < VS; ++s) {
              ao[s] = adjp[(2 * s + h) * 32 + lr];                 // A operand: A^[u = 2s+h][v = lane]
              xo[s] = xrs[2 * s + h];          // B operand: x[c = lane][t][u = 2s+h]; u = V (odd V) reads a finite
                                               // neighbour or the zeroed pad, and A^[u >= V] = 0
            }
#pragma unroll
            for (int s = 0; s < VS; ++s) d = mfma32(ao[s], xo[s], d);
          } else {
            // G^T = A^_sub^T . x^T: A operand = adjacency planes (row v = lane), B operand = this lane's channel, joints
            // 16 ks + 8 h + e (beyond V: finite neighbours that meet the zero rows of the adjacency)
            const unsigned char* aq = dyi + BUF_BYTES - ADJ_BYTES + sub * PLA * 2048 + (h * 32 + lr) * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              u32x4 qh, qm, ql;
#pragma unroll
              for (int e2 = 0; e2 < 4; ++e2) {
                unsigned q0, q1, q2 = 0;
                const float xa = xrs[16 * ks + 8 * h + 2 * e2], xb2 = xrs[16 * ks + 8 * h + 2 * e2 + 1];
                if constexpr (F16) split_pair_f16_mix(xa * s_x, xb2 * s_x, q0, q1);
                else wc_split_pair(xa, xb2, q0, q1, q2);
                qh[e2] = q0; qm[e2] = q1; ql[e2] = q2;
              }
              if constexpr (F16) {
                const f16x8 x0 = __builtin_bit_cast(f16x8, qh), x1 = __builtin_bit_cast(f16x8, qm);
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(aq + 0 * 2048 + ks * 1024);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(aq + 1 * 2048 + ks * 1024);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, x0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, x1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, x0, d, 0, 0, 0);
              } else {
                const bf16x8 x0 = __builtin_bit_cast(bf16x8, qh), x1 = __builtin_bit_cast(bf16x8, qm),
                             x2 = __builtin_bit_cast(bf16x8, ql);
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(aq + 0 * 2048 + ks * 1024);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(aq + 1 * 2048 + ks * 1024);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(aq + 2 * 2048 + ks * 1024);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x2, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x0, d, 0, 0, 0);
              }
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int v = (j & 3) + 8 * (j >> 2) + 4 * h;
            const float t = xr[v];             // (beyond V: a neighbour's finite value, dropped by the select)
            d[j] = (v < V) ? t : 0.f;
          }
        }
        // ---- split and contract with dy: acc[sub][tm][o][c] += sum_v dy[o][v] * G^T[v][c] ----
        bf16x8 gb[2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 gh, gm, gl;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            unsigned q0, q1, q2 = 0;
            if constexpr (F16 && BCH) split_pair_f16_mix(d[8 * ks + 2 * e2], d[8 * ks + 2 * e2 + 1], q0, q1);   // (G carries s_x)
            else if constexpr (F16) split_pair_f16(d[8 * ks + 2 * e2] * s_x, d[8 * ks + 2 * e2 + 1] * s_x, q0, q1);
            else wc_split_pair(d[8 * ks + 2 * e2], d[8 * ks + 2 * e2 + 1], q0, q1, q2);
            gh[e2] = q0; gm[e2] = q1; gl[e2] = q2;
          }
          gb[ks][0] = __builtin_bit_cast(bf16x8, gh);
          gb[ks][1] = __builtin_bit_cast(bf16x8, gm);
          gb[ks][2] = __builtin_bit_cast(bf16x8, gl);
        }
        // 2*TM steps of 6 (f16x3: 3) MFMAs; the dy fragments of step s+1 are read while step s runs (order pinned below)
        auto load_a = [&](bf16x8 (&af)[3], int step) __attribute__((always_inline)) {
          const int ks = step / TM, tm = step - ks * TM;
#pragma unroll
          for (int pl = 0; pl < PL; ++pl)
            af[pl] = *reinterpret_cast<const bf16x8*>(ab + ((pl * FT * 4 + ks * 2) * BMP + tm * 32) * 16);
        };
        bf16x8 afc[3];
        load_a(afc, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, PL, 0);
#pragma unroll
        for (int step = 0; step < 2 * TM; ++step) {
          const int ks = step / TM, tm = step - ks * TM;
          bf16x8 afn[3];
          if (step + 1 < 2 * TM) load_a(afn, step + 1);
          if constexpr (F16) {
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afc[1]), __builtin_bit_cast(f16x8, gb[ks][0]), acc[sub][tm], 0, 0, 0);
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afc[0]), __builtin_bit_cast(f16x8, gb[ks][1]), acc[sub][tm], 0, 0, 0);
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afc[0]), __builtin_bit_cast(f16x8, gb[ks][0]), acc[sub][tm], 0, 0, 0);
          } else {
          if (a.npl == 3) {
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[2], gb[ks][0], acc[sub][tm], 0, 0, 0);
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[0], gb[ks][2], acc[sub][tm], 0, 0, 0);
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[1], gb[ks][1], acc[sub][tm], 0, 0, 0);
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[1], gb[ks][0], acc[sub][tm], 0, 0, 0);
            acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[0], gb[ks][1], acc[sub][tm], 0, 0, 0);
          }
          acc[sub][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[0], gb[ks][0], acc[sub][tm], 0, 0, 0);
          }
          if (step + 1 < 2 * TM) {
            __builtin_amdgcn_sched_group_barrier(0x100, PL, 0);
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) afc[pl] = afn[pl];
          }
          __builtin_amdgcn_sched_group_barrier(0x008, NPROD, 0);
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the subsets apart (interleaving them costs registers)
      }
    }
    stage_barrier();
  }
  // ---- this frame group's partial slab, [z][m][c] (lanes = consecutive c: coalesced rows) ----
  const int slab = (int)blockIdx.y * NFG + fg;
#pragma unroll
  for (int sub = 0; sub < NS; ++sub) {
    float* dst = a.part + (long)slab * a.wsize + (long)sub * a.M * a.C;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int m = m0 + (rb * TM + tm) * 32 + mfma_row(j, h);
        const int c = c0 + cbw * 32 + lr;
        if (m < a.M && c < a.C) dst[(long)m * a.C + c] = F16 ? acc[sub][tm][j] * (inv_dy * inv_x) : acc[sub][tm][j];
      }
  }
}

struct WcGeom {
  int ntiles, ncg, nmb, nsplit, nslabs, pairs_per_split, grid_x, XP;
  size_t smem_bytes;
};

template <int AGG, int TM, int NCB, int NW = 8>
WcGeom wc_geom(int N, int M, int C, int V, int T_out) {
  constexpr int BM = TM * 32, NFG = NW / NCB, FPW = (NCB <= 2) ? 1 : 2, FT = NFG * FPW, CG = NCB * 32;
  WcGeom g;
  g.ntiles = (T_out + FT - 1) / FT;
  g.ncg = (C + CG - 1) / CG;
  g.nmb = (M + BM - 1) / BM;
  g.XP = (FT * V) | 1;
  g.smem_bytes = (size_t)3 * FT * 4 * (BM + 1) * 16 + (size_t)CG * g.XP * 4 + (AGG ? 32 * 32 * 4 : 0);   // sized for pad = 1
  g.grid_x = g.nmb * g.ncg * (AGG ? 3 : 1);
  const int pairs = N * g.ntiles;
  int want = (256 * 8 / NW) / g.grid_x;  // one 8-wave (or two 4-wave) workgroups per CU
  if (want < 1) want = 1;
  if (want > pairs) want = pairs;
  g.pairs_per_split = (pairs + want - 1) / want;
  g.nsplit = (pairs + g.pairs_per_split - 1) / g.pairs_per_split;
  g.nslabs = g.nsplit * NFG;
  return g;
}

template <int AGG, int TM, int NCB, int VS, int NW = 8>
int wc_launch(WcArgs a, void* ws, size_t ws_bytes, int* nslabs_out, hipStream_t stream) {
  const WcGeom g = wc_geom<AGG, TM, NCB, NW>(a.N, a.M, a.C, a.V, a.T_out);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if ((size_t)g.nslabs * a.wsize * 4 > ws_bytes) return AGCN_ERR_WORKSPACE;
  a.part = (float*)ws;
  a.ntiles = g.ntiles; a.pairs_per_split = g.pairs_per_split; a.ncg = g.ncg; a.XP = g.XP;
  auto kern = wgrad_chain_kernel<AGG, TM, NCB, VS, NW>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};   // per (kernel instantiation, device): the attribute is per device
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  hipLaunchKernelGGL(kern, dim3(g.grid_x, g.nsplit), dim3(NW * 64), g.smem_bytes, stream, a);
  *nslabs_out = g.nslabs;
  return agcn_check_launch();
}

template <int AGG, int TM, int NCB>
int wc_dispatch_vs(const WcArgs& a, void* ws, size_t ws_bytes, int* nslabs, hipStream_t s) {
  const int vs = (a.V + 1) / 2;
  if constexpr (!AGG) {
    return wc_launch<AGG, TM, NCB, 1>(a, ws, ws_bytes, nslabs, s);
  } else {
    if (vs == 13) return wc_launch<AGG, TM, NCB, 13>(a, ws, ws_bytes, nslabs, s);
    if (vs == 9) return wc_launch<AGG, TM, NCB, 9>(a, ws, ws_bytes, nslabs, s);
    return wc_launch<AGG, TM, NCB, 16>(a, ws, ws_bytes, nslabs, s);
  }
}

static inline bool wc_four_waves() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AGCN_WGRAD_NW");
    v = (e && atoi(e) == 8) ? 0 : 1;
  }
  return v == 1;
}

template <int AGG, int TM, int NCB, int NW>
int wc_dispatch_vs4(const WcArgs& a, void* ws, size_t ws_bytes, int* nslabs, hipStream_t s) {
  const int vs = (a.V + 1) / 2;
  if constexpr (!AGG) {
    return wc_launch<AGG, TM, NCB, 1, NW>(a, ws, ws_bytes, nslabs, s);
  } else {
    if (vs == 13) return wc_launch<AGG, TM, NCB, 13, NW>(a, ws, ws_bytes, nslabs, s);
    if (vs == 9) return wc_launch<AGG, TM, NCB, 9, NW>(a, ws, ws_bytes, nslabs, s);
    return wc_launch<AGG, TM, NCB, 16, NW>(a, ws, ws_bytes, nslabs, s);
  }
}

template <int AGG, int TM>
int wc_dispatch_ncb(const WcArgs& a, void* ws, size_t ws_bytes, int* nslabs, hipStream_t s) {
  // (the aggregated variant at 128+ channels runs out of registers with 4-wave workgroups: measured 8 % slower)
  if (wc_four_waves() && !AGG && a.C % 128 == 0) return wc_dispatch_vs4<AGG, TM, 4, 4>(a, ws, ws_bytes, nslabs, s);
  if (wc_four_waves() && a.C % 64 == 0 && !(AGG && a.C % 128 == 0)) return wc_dispatch_vs4<AGG, TM, 2, 4>(a, ws, ws_bytes, nslabs, s);
  if (a.C % 256 == 0) return wc_dispatch_vs<AGG, TM, 8>(a, ws, ws_bytes, nslabs, s);
  if (a.C % 128 == 0) return wc_dispatch_vs<AGG, TM, 4>(a, ws, ws_bytes, nslabs, s);
  if (a.C % 64 == 0) return wc_dispatch_vs<AGG, TM, 2>(a, ws, ws_bytes, nslabs, s);
  return wc_dispatch_vs<AGG, TM, 1>(a, ws, ws_bytes, nslabs, s);     // few channels (first layer): one zero-padded block
}

template <int AGG>
size_t wc_slabs(int N, int M, int C, int V, int T_out) {
  const bool tm4 = M > 64 && C % 64 == 0;     // the single-block (few channels) variant stages 8 frames: 64 rows only
  int n;
  if (wc_four_waves() && !AGG && C % 128 == 0) n = tm4 ? wc_geom<AGG, 4, 4, 4>(N, M, C, V, T_out).nslabs : wc_geom<AGG, 2, 4, 4>(N, M, C, V, T_out).nslabs;
  else if (wc_four_waves() && C % 64 == 0 && !(AGG && C % 128 == 0)) n = tm4 ? wc_geom<AGG, 4, 2, 4>(N, M, C, V, T_out).nslabs : wc_geom<AGG, 2, 2, 4>(N, M, C, V, T_out).nslabs;
  else if (C % 256 == 0) n = tm4 ? wc_geom<AGG, 4, 8>(N, M, C, V, T_out).nslabs : wc_geom<AGG, 2, 8>(N, M, C, V, T_out).nslabs;
  else if (C % 128 == 0) n = tm4 ? wc_geom<AGG, 4, 4>(N, M, C, V, T_out).nslabs : wc_geom<AGG, 2, 4>(N, M, C, V, T_out).nslabs;
  else if (C % 64 == 0) n = tm4 ? wc_geom<AGG, 4, 2>(N, M, C, V, T_out).nslabs : wc_geom<AGG, 2, 2>(N, M, C, V, T_out).nslabs;
  else n = tm4 ? wc_geom<AGG, 4, 1>(N, M, C, V, T_out).nslabs : wc_geom<AGG, 2, 1>(N, M, C, V, T_out).nslabs;
  return (size_t)n;
}


// ---- producer / consumer variant: geometry, launch, dispatch ----
template <int AGG, int TM, int NCB, int NRB>
WcGeom wc_geom_pc(int N, int M, int C, int V, int T_out, bool bch = false, int planes = 3) {
  constexpr int BM = TM * 32 * NRB, NFG = 8 / (NCB * NRB), FT = NFG, CG = NCB * 32;
  WcGeom g;
  g.ntiles = (T_out + FT - 1) / FT;
  g.ncg = (C + CG - 1) / CG;
  g.nmb = (M + BM - 1) / BM;
  g.XP = (FT * V) | 1;
  g.smem_bytes = 2 * ((size_t)planes * FT * 4 * (BM + 1) * 16 + (((size_t)CG * g.XP * 4 + 15) & ~(size_t)15) +
                      (AGG ? (bch ? (size_t)3 * planes * 2 * 2 * 32 * 16 + 128 : (size_t)3 * 32 * 32 * 4) : 0)) + 32;
  g.grid_x = g.nmb * g.ncg;
  const int pairs = N * g.ntiles;
  int want = 256 / g.grid_x;                   // one 12-wave workgroup per CU
  if (want < 1) want = 1;
  if (want > pairs) want = pairs;
  g.pairs_per_split = (pairs + want - 1) / want;
  g.nsplit = (pairs + g.pairs_per_split - 1) / g.pairs_per_split;
  g.nslabs = g.nsplit * NFG;
  return g;
}

// AGCN_WGRAD_F16X3=0 keeps the weight gradients on bf16x6 (A/B)
static inline bool wc_f16x3() {
  static const int on = getenv("AGCN_WGRAD_F16X3") ? atoi(getenv("AGCN_WGRAD_F16X3")) : 1;
  return on != 0;
}

template <int AGG, int TM, int NCB, int NRB, int VS, int NWP = 4>
int wc_launch_pc(WcArgs a, void* ws, size_t ws_bytes, int* nslabs_out, hipStream_t stream) {
  // f16x3 when the caller supplied both operand maxima (fp32-equivalent mode); VS == 0: the aggregation on f16x3 too
  {
    if (a.dy_absmax && a.in_absmax && a.npl == 3 && wc_f16x3()) {
      const WcGeom g = wc_geom_pc<AGG, TM, NCB, NRB>(a.N, a.M, a.C, a.V, a.T_out, AGG && VS == 0, 2);
      if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
      if ((size_t)g.nslabs * a.wsize * 4 > ws_bytes) return AGCN_ERR_WORKSPACE;
      a.part = (float*)ws;
      a.ntiles = g.ntiles; a.pairs_per_split = g.pairs_per_split; a.ncg = g.ncg; a.XP = g.XP;
      auto kern = wgrad_pc_kernel<AGG, TM, NCB, VS, NRB, 8, NWP, true>;
      static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};
      if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
      AGCN_NOTE_KERNEL("wgrad_pc_kernel<%d, %d, %d, %d, %d, 8, %d, true>", AGG, TM, NCB, VS, NRB, NWP);
      hipLaunchKernelGGL(kern, dim3(g.grid_x, g.nsplit), dim3((8 + NWP) * 64), g.smem_bytes, stream, a);
      *nslabs_out = g.nslabs;
      return agcn_check_launch();
    }
  }
  const WcGeom g = wc_geom_pc<AGG, TM, NCB, NRB>(a.N, a.M, a.C, a.V, a.T_out, AGG && VS == 0);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if ((size_t)g.nslabs * a.wsize * 4 > ws_bytes) return AGCN_ERR_WORKSPACE;
  a.part = (float*)ws;
  a.ntiles = g.ntiles; a.pairs_per_split = g.pairs_per_split; a.ncg = g.ncg; a.XP = g.XP;
  auto kern = wgrad_pc_kernel<AGG, TM, NCB, VS, NRB, 8, NWP>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  AGCN_NOTE_KERNEL("wgrad_pc_kernel<%d, %d, %d, %d, %d, 8, %d>", AGG, TM, NCB, VS, NRB, NWP);
  hipLaunchKernelGGL(kern, dim3(g.grid_x, g.nsplit), dim3((8 + NWP) * 64), g.smem_bytes, stream, a);
  *nslabs_out = g.nslabs;
  return agcn_check_launch();
}

template <int AGG, int TM, int NCB, int NRB>
int wc_dispatch_pc_vs(const WcArgs& a, void* ws, size_t ws_bytes, int* nslabs, hipStream_t s) {
  const int vs = (a.V + 1) / 2;
  if constexpr (!AGG) {
    return wc_launch_pc<AGG, TM, NCB, NRB, 1>(a, ws, ws_bytes, nslabs, s);
  } else {
    // (the split-bf16 aggregation measured 3-7 % SLOWER here: the consumer re-splits its x fragment for every subset
    // and becomes VALU-bound; AGCN_WC_BCH=1 selects it)
    static const int bch = getenv("AGCN_WC_BCH") ? atoi(getenv("AGCN_WC_BCH")) : 0;
    if (bch && a.npl == 3) return wc_launch_pc<AGG, TM, NCB, NRB, 0>(a, ws, ws_bytes, nslabs, s);
    // with both maxima: the aggregation on f16x3 as well (AGCN_WC_AGG16=0: exact-f32 aggregation chain, round 2)
    static const int agg16 = getenv("AGCN_WC_AGG16") ? atoi(getenv("AGCN_WC_AGG16")) : 1;
    if (agg16 && a.dy_absmax && a.in_absmax && a.npl == 3 && wc_f16x3())
      return wc_launch_pc<AGG, TM, NCB, NRB, 0>(a, ws, ws_bytes, nslabs, s);
    if (vs == 13) return wc_launch_pc<AGG, TM, NCB, NRB, 13>(a, ws, ws_bytes, nslabs, s);
    if (vs == 9) return wc_launch_pc<AGG, TM, NCB, NRB, 9>(a, ws, ws_bytes, nslabs, s);
    return wc_launch_pc<AGG, TM, NCB, NRB, 16>(a, ws, ws_bytes, nslabs, s);
  }
}

// AGCN_WC_PC=0 selects the single-role kernel (A/B measurement)
static inline bool wc_pc_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AGCN_WC_PC");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v == 1;
}
static inline bool wc_pc_applies(int C) { return wc_pc_enabled() && C % 64 == 0; }

// Tiles (8 consumer waves, 2 frames per stage).  AGG: a consumer wave holds 3 subsets x 2 row tiles of accumulators:
// 128 channels x 64 rows, or 64 channels x 128 rows.  Plain: 128 channels x 128 rows (4 row tiles per wave), or
// 64 channels x 2 row blocks of 128.
template <int AGG>
int wc_dispatch_pc(const WcArgs& a, void* ws, size_t ws_bytes, int* nslabs, hipStream_t s) {
  if constexpr (AGG) {
    if (a.C % 128 == 0) return wc_dispatch_pc_vs<1, 2, 4, 1>(a, ws, ws_bytes, nslabs, s);
    return wc_dispatch_pc_vs<1, 2, 2, 2>(a, ws, ws_bytes, nslabs, s);
  } else {
    // The plain 1x1 gradients are producer-bound (both operands split by the 4 producers).  Up to 192 rows, 8 producer
    // waves on 64-row blocks (4 waves per SIMD, 128 VGPRs) win: l6 shape 390 -> 322 us; with more rows the extra
    // passes over x cost more than the producers gain (l9: 628 -> 697 us).  AGCN_WC_NWP8=0 / 1 forces either.
    static const int nwp8 = getenv("AGCN_WC_NWP8") ? atoi(getenv("AGCN_WC_NWP8")) : -1;
    if (a.C % 128 == 0 && (nwp8 == 1 || (nwp8 < 0 && a.M > 64 && a.M <= 192)))
      return wc_launch_pc<0, 2, 4, 1, 1, 8>(a, ws, ws_bytes, nslabs, s);
    if (a.C % 128 == 0) {
      if (a.M > 64) return wc_dispatch_pc_vs<0, 4, 4, 1>(a, ws, ws_bytes, nslabs, s);
      return wc_dispatch_pc_vs<0, 2, 4, 1>(a, ws, ws_bytes, nslabs, s);
    }
    return wc_dispatch_pc_vs<0, 2, 2, 2>(a, ws, ws_bytes, nslabs, s);
  }
}

template <int AGG>
size_t wc_slabs_pc(int N, int M, int C, int V, int T_out) {
  if constexpr (AGG) {
    if (C % 128 == 0) return (size_t)wc_geom_pc<1, 2, 4, 1>(N, M, C, V, T_out).nslabs;
    return (size_t)wc_geom_pc<1, 2, 2, 2>(N, M, C, V, T_out).nslabs;
  } else {
    if (C % 128 == 0) {
      const size_t n4 = (size_t)wc_geom_pc<0, 4, 4, 1>(N, M, C, V, T_out).nslabs;
      const size_t n2 = (size_t)wc_geom_pc<0, 2, 4, 1>(N, M, C, V, T_out).nslabs;
      return n4 > n2 ? n4 : n2;
    }
    return (size_t)wc_geom_pc<0, 2, 2, 2>(N, M, C, V, T_out).nslabs;
  }
}

}  // namespace

// C a multiple of 64, or at most 32 (one zero-padded channel block)
bool agcn_wgrad_chain_supported(int M, int C, int V) { return M >= 64 && (C % 64 == 0 || C <= 32) && C >= 1 && V <= 32; }

size_t agcn_wgrad_chain_workspace(int agg, int N, int M, int C, int V, int T_out) {
  const long wsize = (long)(agg ? 3 : 1) * M * C;
  size_t n = agg ? wc_slabs<1>(N, M, C, V, T_out) : wc_slabs<0>(N, M, C, V, T_out);
  if (C % 64 == 0) {                            // either variant may run (AGCN_WC_PC): size for the larger
    const size_t m = agg ? wc_slabs_pc<1>(N, M, C, V, T_out) : wc_slabs_pc<0>(N, M, C, V, T_out);
    if (m > n) n = m;
  }
  return n * (size_t)wsize * 4;
}

// writes the partial slabs into ws; *nslabs = number of slabs for the reduce kernel.  agg: x . adj_i operand, z = subset
int agcn_wgrad_chain(int agg, const float* dy, const float* x, const float* adj, void* ws, size_t ws_bytes, int* nslabs,
                     int N, int M, int C, int V, int T_src, int T_out, int stride, hipStream_t s, const float* dy_absmax,
                     const float* x_absmax) {
  WcArgs a = {};
  a.dy_absmax = dy_absmax; a.in_absmax = x_absmax;
  a.npl = agcn_npl();
  { const char* e = getenv("AGCN_WC_PAD"); a.pad = (e && atoi(e) == 0) ? 0 : 1; }
  { const char* e = getenv("AGCN_WC_DBG"); a.dbg = e ? atoi(e) : 0; }
  a.dy = dy; a.in = x; a.adj = adj; a.N = N; a.M = M; a.C = C; a.V = V; a.T_src = T_src; a.T_out = T_out;
  a.stride = stride; a.wsize = (long)(agg ? 3 : 1) * M * C;
  // (64 channels x 64 rows with aggregation: the two-row-block tile of the producer/consumer kernel would be half empty)
  if (wc_pc_applies(C) && stride == 1 && !(agg && C % 128 != 0 && M <= 64)) return agg ? wc_dispatch_pc<1>(a, ws, ws_bytes, nslabs, s) : wc_dispatch_pc<0>(a, ws, ws_bytes, nslabs, s);
  const bool tm4 = M > 64 && C % 64 == 0;
  if (agg) return tm4 ? wc_dispatch_ncb<1, 4>(a, ws, ws_bytes, nslabs, s) : wc_dispatch_ncb<1, 2>(a, ws, ws_bytes, nslabs, s);
  return tm4 ? wc_dispatch_ncb<0, 4>(a, ws, ws_bytes, nslabs, s) : wc_dispatch_ncb<0, 2>(a, ws, ws_bytes, nslabs, s);
}
