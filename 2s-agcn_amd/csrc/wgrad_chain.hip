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

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
  int pad;             // 1: [o] blocks of the dy image padded by one slot (AGCN_WC_PAD=0 for the A/B measurement)
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
  auto issue = [&](int p) __attribute__((always_inline)) {
    int n, t0;
    pair_geom(p, n, t0);
#pragma unroll
    for (int k = 0; k < DI; ++k) {
      const int item = tid + k * NT;                 // = o*(FT*4) + f*4 + (ks*2 + hh)
      const int o = item / (FT * 4), r = item - o * (FT * 4);
      const int f = r >> 2, ks = (r >> 1) & 1, hh = r & 1;
      const bool ok = (m0 + o) < a.M && (t0 + f) < a.T_out;
      const float* src = a.dy + ((long)n * a.M + (ok ? (m0 + o) : 0)) * Pout + (long)(ok ? (t0 + f) : 0) * V;
#pragma unroll
      for (int e = 0; e < 8; ++e) rdy[k][e] = src[min(wc_joint(ks, hh, e), V - 1)];
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
    commit(p);
    int n, t0;
    pair_geom(p, n, t0);
    if (AGG && n != last_n) {
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
    if (p + 1 < p_end) issue(p + 1);          // in flight during this pair's matrix-core work
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      const int f = fg + k * NFG;
      if (t0 + f >= a.T_out) continue;        // wave-uniform
      // ---- G^T[v][c] for (frame f, this wave's 32 channels) in D layout ----
      f32x16 d;
      const float* xr = xs + (cbw * 32 + lr) * XP + f * V;
      if (AGG) {
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
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 gh, gm, gl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          unsigned q0, q1, q2;
          wc_split_pair(d[8 * ks + 2 * e2], d[8 * ks + 2 * e2 + 1], q0, q1, q2);
          gh[e2] = q0; gm[e2] = q1; gl[e2] = q2;
        }
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, gh), b1 = __builtin_bit_cast(bf16x8, gm),
                     b2 = __builtin_bit_cast(bf16x8, gl);
        const unsigned char* ab = dyi + ((((f * 2 + ks) * 2 + h) * BMP) + lr) * 16;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ab + ((0 * FT * 4) * BMP + tm * 32) * 16);
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ab + ((1 * FT * 4) * BMP + tm * 32) * 16);
          const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ab + ((2 * FT * 4) * BMP + tm * 32) * 16);
          if (a.npl == 3) {
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[tm], 0, 0, 0);
          }
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[tm], 0, 0, 0);
        }
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

}  // namespace

// C a multiple of 64, or at most 32 (one zero-padded channel block)
bool agcn_wgrad_chain_supported(int M, int C, int V) { return M >= 64 && (C % 64 == 0 || C <= 32) && C >= 1 && V <= 32; }

size_t agcn_wgrad_chain_workspace(int agg, int N, int M, int C, int V, int T_out) {
  const long wsize = (long)(agg ? 3 : 1) * M * C;
  return (agg ? wc_slabs<1>(N, M, C, V, T_out) : wc_slabs<0>(N, M, C, V, T_out)) * (size_t)wsize * 4;
}

// writes the partial slabs into ws; *nslabs = number of slabs for the reduce kernel.  agg: x . adj_i operand, z = subset
int agcn_wgrad_chain(int agg, const float* dy, const float* x, const float* adj, void* ws, size_t ws_bytes, int* nslabs,
                     int N, int M, int C, int V, int T_src, int T_out, int stride, hipStream_t s) {
  WcArgs a = {};
  a.npl = agcn_npl();
  { const char* e = getenv("AGCN_WC_PAD"); a.pad = (e && atoi(e) == 0) ? 0 : 1; }
  a.dy = dy; a.in = x; a.adj = adj; a.N = N; a.M = M; a.C = C; a.V = V; a.T_src = T_src; a.T_out = T_out;
  a.stride = stride; a.wsize = (long)(agg ? 3 : 1) * M * C;
  const bool tm4 = M > 64 && C % 64 == 0;
  if (agg) return tm4 ? wc_dispatch_ncb<1, 4>(a, ws, ws_bytes, nslabs, s) : wc_dispatch_ncb<1, 2>(a, ws, ws_bytes, nslabs, s);
  return tm4 ? wc_dispatch_ncb<0, 4>(a, ws, ws_bytes, nslabs, s) : wc_dispatch_ncb<0, 2>(a, ws, ws_bytes, nslabs, s);
}
