// Weight gradient of unit_tcn's 9x1 temporal convolution (reference agcn.py:40-41 differentiated) in split-bf16
// ("bf16x6", fp32-equivalent) arithmetic:
//     dW[m][c][k] = sum_{n,t,v} dy[n][m][t][v] * x[n][c][t + k - 4][v]                      (stride 1)
// GEMM view: rows m, columns (c, k), contraction over positions.  The nine taps of one channel read the SAME input
// shifted along t, so the contraction axis is laid out t-contiguous: for one (sample, joint) a K-step is 16 consecutive
// frames, the A fragment of a lane is 8 consecutive frames of a dy row, and ONE 16-frame window of an x row holds the
// B fragments of all nine taps -- a tap is a shift of the window by `tap` elements (even shifts are register renames,
// odd shifts one v_alignbit per register).  6 wide LDS reads feed 54 MFMAs (9 taps x 6 split products).
//
// The activations live in HBM as (N, C, T, V) with v fastest, so a small transposing pre-pass writes t-contiguous
// copies (N, V, C, Tp) of dy and x (Tp = T rounded up to 4, zero padded) -- 2 x (read + write) of the two tensors,
// ~0.16 ms per layer at the l2-l4 size against ~1.3 ms saved -- and the main kernel streams whole rows with float4 loads.
// The dy tile is kept in LDS as fp32 and split into the three bf16 pieces in registers (one split per 54 MFMAs, 4 bytes
// of LDS per element); the x rows, which every row-tile wave of a channel tile reads, are split once while staged and
// kept as three bf16 planes.
//
// Workgroup = 4 waves = (128 rows x 32 channels) or (64 x 64), all 9 taps; wave = one 32x32 (row tile, channel tile) for
// all taps (144 accumulator registers); two workgroups per CU; split-K over (sample, joint, 80-frame chunk) units, one slab per split, summed in a
// fixed order by the caller's slab reduction (bitwise reproducible).
#include "agcn_common.h"
#include "split_bf16.h"
#include "split_f16.h"

namespace {

constexpr int TC = 80;        // output frames per chunk = 5 K-steps of 16
constexpr int DP = 84;        // LDS pitch of a dy row (floats): 4 * odd -> conflict-free 16-byte reads across rows
constexpr int XW = TC + 8;    // frames of an x row a chunk needs: t0-4 .. t0+TC+3
constexpr int XPB = 88;       // LDS pitch of an x row of one bf16 plane (elements): 176 bytes = 16 * odd

struct TrArgs {
  const float* in;   // (N, R, T, V)
  float* out;        // (N, V, R, S, Tp): frame t goes to parity row t % S, position t / S  (S = 1: plain t-contiguous rows)
  int N, R, T, V, Tp, S;
  const float* absmax;   // F16: device scalar max |in| (range scale); the output is two fp16 planes of `plane` elements each
  long plane;
};

// (N, R, T, V) -> (N, V, R, Tp): t-contiguous rows, [T, Tp) zero.  A workgroup moves 8 rows x 64 frames through LDS:
// the source tile (8 contiguous runs of 64*V floats) comes in by direct-to-LDS loads (no registers, every load of the
// tile in flight at once: ~50 KB per workgroup, three workgroups per CU), and leaves as 16-byte stores of 4 frames.
// F16: the copy is written as the two fp16 planes of the f16x3 split (hi, then residual; same bytes as the fp32 copy),
// range-scaled by max |in|: the main kernel then loads matrix operands as they are, with no conversion work left in it.
// The F16 variant moves RB = 4 rows x TT = 128 frames per workgroup and a lane writes FPL = 8 frames: 16-byte stores, 256
// contiguous bytes per (joint, row, plane) instead of 128.
template <bool F16, int RB, int TT, int FPL>
__global__ void __launch_bounds__(256) tv_transpose_kernel(const TrArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  static_assert(FPL == 4 || (F16 && FPL == 8), "a lane writes one 16-byte store per plane");
  constexpr int QL = TT / FPL, NSUB = 64 / QL;   // lanes per (joint, row) pair, pairs per wave-instruction
  const int S = a.S;
  const int ntt = (a.Tp * S + TT - 1) / TT, nrb = (a.R + RB - 1) / RB;
  int b = blockIdx.x;
  const int tti = b % ntt;
  b /= ntt;
  const int rb = b % nrb, n = b / nrb;
  const int t0 = tti * TT, r0 = rb * RB;
  const int tl = max(0, min(TT, a.T - t0));      // source frames of this tile
  const int len = tl * a.V;                      // floats per row
  const int pitch = TT * a.V + 4;                // row pitch in LDS (a load instruction never crosses a row)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // 16 bytes per lane (1 KB per wave-instruction) where the source rows allow it: T*V and the tile's length multiples of 4
  // floats and a 16-byte aligned tensor (workgroup-uniform); 4 bytes per lane otherwise
  const bool wide = ((a.T * a.V) & 3) == 0 && (len & 3) == 0 && len > 0 && ((reinterpret_cast<unsigned long>(a.in) & 15) == 0);
  if (wide) {
    const int npiece = (len + 255) >> 8;         // 256-float pieces per row
    for (int p = wave; p < RB * npiece; p += 4) {
      const int r = p / npiece, e0 = (p - r * npiece) * 256;
      if (r0 + r < a.R) {
        const float* src = a.in + (((long)n * a.R + r0 + r) * a.T + t0) * a.V + e0;
        // (lanes past the row's end stay out: the destination of lane l is base + 16 l, and the next row starts there)
        if (e0 + 4 * lane < len)
          __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * lane), (lptr_t)(tile + r * pitch + e0), 16, 0, 0);
      }
    }
  } else {
    const int npiece = (len + 63) >> 6;          // 64-float pieces per row
    for (int p = wave; p < RB * npiece; p += 4) {  // wave-uniform piece -> (row, offset)
      const int r = p / npiece, e0 = (p - r * npiece) * 64;
      if (r0 + r < a.R) {
        const float* src = a.in + (((long)n * a.R + r0 + r) * a.T + t0) * a.V + e0;
        const int e = min(lane, len - 1 - e0);   // the ragged last piece re-reads its last element (never used)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + e), (lptr_t)(tile + r * pitch + e0), 4, 0, 0);
      }
    }
  }
  __syncthreads();                               // (drains the LDS-DMA: the compiler waits vmcnt(0) here)
  float rs_s = 1.f, rs_inv = 1.f;
  if constexpr (F16) f16_range_scale(a.absmax, rs_s, rs_inv);
  (void)rs_inv;
  const int TTo = TT / S;                        // output positions per parity row in this tile
  const int tw = min(TTo, a.Tp - t0 / S);        // positions written per parity row (zero pad included); multiple of FPL
  const int qp = TTo / FPL;                      // lanes per parity row
  const int lq = lane % QL;
  const int par = lq / qp, q = lq - par * qp, sub = lane / QL;
  for (int p = wave * NSUB + sub; p < a.V * RB; p += 4 * NSUB) {   // lane -> (parity, group), NSUB (joint, row) pairs per instruction
    const int v = p / RB, r = p - v * RB;
    if (r0 + r < a.R && FPL * q < tw) {
      float val[FPL];
#pragma unroll
      for (int j = 0; j < FPL; ++j) {
        const int f = (FPL * q + j) * S + par;   // source frame within the tile
        val[j] = (f < tl) ? tile[r * pitch + f * a.V + v] : 0.f;
      }
      const long o = ((((long)n * a.V + v) * a.R + r0 + r) * S + par) * a.Tp + t0 / S + FPL * q;
      if constexpr (F16) {
        unsigned hh[FPL / 2], ll[FPL / 2];
#pragma unroll
        for (int j = 0; j < FPL / 2; ++j) split_pair_f16_mix(val[2 * j] * rs_s, val[2 * j + 1] * rs_s, hh[j], ll[j]);
        unsigned short* oh = reinterpret_cast<unsigned short*>(a.out);
        if constexpr (FPL == 8) {
          *reinterpret_cast<u32x4*>(oh + o) = u32x4{hh[0], hh[1], hh[2], hh[3]};
          *reinterpret_cast<u32x4*>(oh + a.plane + o) = u32x4{ll[0], ll[1], ll[2], ll[3]};
        } else {
          *reinterpret_cast<uint2*>(oh + o) = make_uint2(hh[0], hh[1]);
          *reinterpret_cast<uint2*>(oh + a.plane + o) = make_uint2(ll[0], ll[1]);
        }
      } else {
        *reinterpret_cast<f32x4*>(a.out + o) = f32x4{val[0], val[1], val[2], val[3]};
      }
    }
  }
}

struct W9Args {
  const float* dyT;   // (N, V, M, Tp)
  const float* xT;    // (N, V, C, S, Tp): S parity rows per channel (S = stride of the convolution)
  float* part;        // [nsplit][9][M][C]
  int N, M, C, V, T, Tp;
  int nchunk, units, units_per_split, ncb;
  int npl;            // 3: six split products (bf16x6) ; 1: hi*hi only (bf16)
  // f16x3 variant: the transposed copies are fp16 plane pairs (plane stride in elements), scaled by these maxima
  const unsigned short* dyH;
  const unsigned short* xH;
  long dy_plane, x_plane;
  const float* dy_absmax;
  const float* x_absmax;
};

// elements s .. s+7 of a 16-element window held as 8 packed bf16 pairs
template <int S>
__device__ __forceinline__ bf16x8 window_frag(const unsigned (&p)[8]) {
  u32x4 r;
  if (S % 2 == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = p[S / 2 + i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_alignbit(p[(S + 1) / 2 + i], p[(S - 1) / 2 + i], 16);
  }
  return __builtin_bit_cast(bf16x8, r);
}

template <int S>
__device__ __forceinline__ f32x16 tap_mfma(bf16x8 a0, bf16x8 a1, bf16x8 a2, const unsigned (&p0)[8],
                                           const unsigned (&p1)[8], const unsigned (&p2)[8], f32x16 c, int npl) {
  return sb_mfma6(a0, a1, a2, window_frag<S>(p0), window_frag<S>(p1), window_frag<S>(p2), c, npl);
}

// RT x CT = 4 waves: (4, 1) = 128 rows x 32 channels, (2, 2) = 64 rows x 64 channels.  Two workgroups share a CU
// (<= 256 VGPRs, <= 55 KB of LDS each): a unit's operands are fetched with ALL loads in flight at once and no register
// prefetch across units -- while one workgroup stages, the other one keeps the matrix pipe busy.
// STRIDE 2: x[2t + k - 4] = xE[t + (k-4)/2] for even taps, xO[t + (k-5)/2] for odd taps, where xE / xO are the even- and
// odd-frame rows the transposer wrote: two windows per K-step, and a tap is again a shift of one of them.
template <int RT, int CT, int STRIDE>
__global__ void __launch_bounds__(256, 2) wgrad9_bf16_kernel(const W9Args a) {
  constexpr int NT = 256, BM = RT * 32, CBW = CT * 32, SETS = STRIDE;
  static_assert(RT * CT == 4, "four waves");
  constexpr int ND = BM * (TC / 4) / NT;                 // float4 of the dy tile per thread (exact)
  constexpr int NXG = CBW * SETS * (XW / 8);             // 8-frame groups of the x tile (parity rows count as rows)
  constexpr int NX = (NXG + NT - 1) / NT;
  static_assert(BM * (TC / 4) % NT == 0, "dy tile divides evenly");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dyL = smem;                 // [BM][DP] fp32
  unsigned short* xP = reinterpret_cast<unsigned short*>(smem + BM * DP);   // [3 planes][CBW][XPB] bf16: the x rows are
  //                     split ONCE while they are staged (every wave of a channel tile reads the same window)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int rt = wave % RT, ct = wave / RT;
  const int mb = blockIdx.x / a.ncb, cb = blockIdx.x - mb * a.ncb;
  const int m0 = mb * BM, c0 = cb * CBW;
  const int split = blockIdx.y;
  const int u_begin = split * a.units_per_split;
  const int u_end = min(a.units, u_begin + a.units_per_split);
  const int Tp = a.Tp;

  f32x16 acc[9];
#pragma unroll
  for (int z = 0; z < 9; ++z)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[z][j] = 0.f;

  const float* arow = dyL + (rt * 32 + lr) * DP + 8 * h;
  const unsigned short* brow = xP + (ct * 32 + lr) * XPB + 8 * h;          // + (plane * SETS + set) * CBW * XPB

  for (int unit = u_begin; unit < u_end; ++unit) {
    const int ch = unit % a.nchunk;
    const int nv = unit / a.nchunk;              // n * V + v
    const int t0 = ch * TC;
    const float* src = a.dyT + ((long)nv * a.M + m0) * Tp + t0;
    const float* sx = a.xT + ((long)nv * a.C + c0) * SETS * Tp + (t0 - 4);   // rows: (channel, parity)
    // All loads of the unit in flight at once.  The slot -> (row, quad) arithmetic is redone per unit on a laundered
    // copy of the thread id: left to itself the compiler hoists the 26 loop-invariant offsets out of the unit loop and
    // keeps them alive across the MFMA loop, which already uses ~236 of the 256 registers (it then spills).
    int tv = tid;
    asm volatile("" : "+v"(tv));
    f32x4 dv[ND], xv[NX][2];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tv + u * NT, r = i / (TC / 4), q = i - r * (TC / 4);
      const bool ok = (m0 + r < a.M) && (t0 + 4 * q < Tp);
      dv[u] = *reinterpret_cast<const f32x4*>(src + (ok ? r * Tp + 4 * q : 0));
      if (!ok) dv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int i = min(tv + u * NT, NXG - 1), r = i / (XW / 8), k = i - r * (XW / 8);      // r = channel * SETS + parity
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const int t = t0 - 4 + 8 * k + 4 * hq;
        const bool ok = (c0 + r / SETS < a.C) && t >= 0 && t < Tp;
        xv[u][hq] = *reinterpret_cast<const f32x4*>(sx + (ok ? r * Tp + 8 * k + 4 * hq : 4));
        if (!ok) xv[u][hq] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();                             // the previous unit's fragment reads are done
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tv + u * NT, r = i / (TC / 4), q = i - r * (TC / 4);
      *reinterpret_cast<f32x4*>(dyL + r * DP + 4 * q) = dv[u];
    }
#pragma unroll
    for (int u = 0; u < NX; ++u) {               // (tail threads rewrite the last group with its own value)
      const int i = min(tv + u * NT, NXG - 1), r = i / (XW / 8), k = i - r * (XW / 8);
      u32x4 ph, pm, pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned y0, y1, y2;
        sb_split_pair(xv[u][e >> 1][2 * (e & 1)], xv[u][e >> 1][2 * (e & 1) + 1], y0, y1, y2);
        ph[e] = y0; pm[e] = y1; pl[e] = y2;
      }
      const int row = (r % SETS) * CBW + r / SETS;           // LDS rows: [plane][parity][channel]
      *reinterpret_cast<u32x4*>(xP + (0 * SETS * CBW + row) * XPB + 8 * k) = ph;
      *reinterpret_cast<u32x4*>(xP + (1 * SETS * CBW + row) * XPB + 8 * k) = pm;
      *reinterpret_cast<u32x4*>(xP + (2 * SETS * CBW + row) * XPB + 8 * k) = pl;
    }
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < TC / 16; ++ks) {
      // A fragment: 8 consecutive frames of this lane's dy row ; B window: 16 consecutive frames of its x row
      const f32x4 ar0 = *reinterpret_cast<const f32x4*>(arow + 16 * ks);
      const f32x4 ar1 = *reinterpret_cast<const f32x4*>(arow + 16 * ks + 4);
      u32x4 bw[SETS][3][2];                        // 16-frame window of each plane (of each parity row)
#pragma unroll
      for (int st = 0; st < SETS; ++st)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            bw[st][pl][i] = *reinterpret_cast<const u32x4*>(brow + (pl * SETS + st) * CBW * XPB + 16 * ks + 8 * i);
      u32x4 ah, am, al;
      {
        unsigned x0, x1, x2;
        sb_split_pair(ar0[0], ar0[1], x0, x1, x2); ah[0] = x0; am[0] = x1; al[0] = x2;
        sb_split_pair(ar0[2], ar0[3], x0, x1, x2); ah[1] = x0; am[1] = x1; al[1] = x2;
        sb_split_pair(ar1[0], ar1[1], x0, x1, x2); ah[2] = x0; am[2] = x1; al[2] = x2;
        sb_split_pair(ar1[2], ar1[3], x0, x1, x2); ah[3] = x0; am[3] = x1; al[3] = x2;
      }
      unsigned p0[8], p1[8], p2[8], o0[8], o1[8], o2[8];   // even-row (or only) window, odd-row window
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        p0[i] = bw[0][0][i >> 2][i & 3];
        p1[i] = bw[0][1][i >> 2][i & 3];
        p2[i] = bw[0][2][i >> 2][i & 3];
        o0[i] = bw[SETS - 1][0][i >> 2][i & 3];
        o1[i] = bw[SETS - 1][1][i >> 2][i & 3];
        o2[i] = bw[SETS - 1][2][i >> 2][i & 3];
      }
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah), a1 = __builtin_bit_cast(bf16x8, am),
                   a2 = __builtin_bit_cast(bf16x8, al);
      if (STRIDE == 1) {                           // tap k = window shift k
        acc[0] = tap_mfma<0>(a0, a1, a2, p0, p1, p2, acc[0], a.npl);
        acc[1] = tap_mfma<1>(a0, a1, a2, p0, p1, p2, acc[1], a.npl);
        acc[2] = tap_mfma<2>(a0, a1, a2, p0, p1, p2, acc[2], a.npl);
        acc[3] = tap_mfma<3>(a0, a1, a2, p0, p1, p2, acc[3], a.npl);
        acc[4] = tap_mfma<4>(a0, a1, a2, p0, p1, p2, acc[4], a.npl);
        acc[5] = tap_mfma<5>(a0, a1, a2, p0, p1, p2, acc[5], a.npl);
        acc[6] = tap_mfma<6>(a0, a1, a2, p0, p1, p2, acc[6], a.npl);
        acc[7] = tap_mfma<7>(a0, a1, a2, p0, p1, p2, acc[7], a.npl);
        acc[8] = tap_mfma<8>(a0, a1, a2, p0, p1, p2, acc[8], a.npl);
      } else {                                     // even taps: xE shifted by 4 + (k-4)/2 ; odd taps: xO by 4 + (k-5)/2
        acc[0] = tap_mfma<2>(a0, a1, a2, p0, p1, p2, acc[0], a.npl);
        acc[1] = tap_mfma<2>(a0, a1, a2, o0, o1, o2, acc[1], a.npl);
        acc[2] = tap_mfma<3>(a0, a1, a2, p0, p1, p2, acc[2], a.npl);
        acc[3] = tap_mfma<3>(a0, a1, a2, o0, o1, o2, acc[3], a.npl);
        acc[4] = tap_mfma<4>(a0, a1, a2, p0, p1, p2, acc[4], a.npl);
        acc[5] = tap_mfma<4>(a0, a1, a2, o0, o1, o2, acc[5], a.npl);
        acc[6] = tap_mfma<5>(a0, a1, a2, p0, p1, p2, acc[6], a.npl);
        acc[7] = tap_mfma<5>(a0, a1, a2, o0, o1, o2, acc[7], a.npl);
        acc[8] = tap_mfma<6>(a0, a1, a2, p0, p1, p2, acc[8], a.npl);
      }
    }
  }

  float* dst = a.part + (long)split * 9 * a.M * a.C;
#pragma unroll
  for (int z = 0; z < 9; ++z)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int m = m0 + rt * 32 + mfma_row(j, h);
      const int c = c0 + ct * 32 + lr;
      if (m < a.M && c < a.C) dst[((long)z * a.M + m) * a.C + c] = acc[z][j];
    }
}


// f16x3 variant (split_f16.h): the operands arrive as fp16 plane pairs from the transposer, so a unit is staged with plain
// 8-byte copies and a K-step is 6 wide LDS reads + 27 MFMAs (9 taps x 3 products) + the odd taps' v_alignbit: half the
// matrix work of bf16x6 and none of its conversion work.
constexpr int DPH = 88;       // LDS pitch of a dy row of one fp16 plane (elements): 176 bytes = 16 * odd

template <int S>
__device__ __forceinline__ f32x16 tap_mfma_f16(f16x8 ah, f16x8 al, const unsigned (&ph)[8], const unsigned (&pl)[8],
                                               f32x16 c) {
  const f16x8 bh = __builtin_bit_cast(f16x8, window_frag<S>(ph)), bl = __builtin_bit_cast(f16x8, window_frag<S>(pl));
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  return c;
}

template <int RT, int CT, int STRIDE>
__global__ void __launch_bounds__(256, 2) wgrad9_f16_kernel(const W9Args a) {
  constexpr int NT = 256, BM = RT * 32, CBW = CT * 32, SETS = STRIDE;
  static_assert(RT * CT == 4, "four waves");
  constexpr int ND = BM * (TC / 4) / NT;                 // 4-frame groups of the dy tile per thread and plane (exact)
  constexpr int NXG = CBW * SETS * (XW / 8);             // 8-frame groups of the x tile (parity rows count as rows)
  constexpr int NX = (NXG + NT - 1) / NT;
  static_assert(BM * (TC / 4) % NT == 0, "dy tile divides evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned short smh[];
  unsigned short* dyL = smh;                             // [2 planes][BM][DPH]
  unsigned short* xP = smh + 2 * BM * DPH;               // [2 planes][SETS][CBW][XPB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int rt = wave % RT, ct = wave / RT;
  const int mb = blockIdx.x / a.ncb, cb = blockIdx.x - mb * a.ncb;
  const int m0 = mb * BM, c0 = cb * CBW;
  const int split = blockIdx.y;
  const int u_begin = split * a.units_per_split;
  const int u_end = min(a.units, u_begin + a.units_per_split);
  const int Tp = a.Tp;

  f32x16 acc[9];
#pragma unroll
  for (int z = 0; z < 9; ++z)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[z][j] = 0.f;

  const unsigned short* arow = dyL + (rt * 32 + lr) * DPH + 8 * h;            // + plane * BM * DPH
  const unsigned short* brow = xP + (ct * 32 + lr) * XPB + 8 * h;             // + (plane * SETS + set) * CBW * XPB

  // A unit's operands are fetched into registers ONE UNIT AHEAD (PREF; all loads in flight at once, nothing consumes them
  // until the next unit's staging), so that their latency runs under this unit's 135 matrix operations; the stride-2
  // variant has no registers left for that and relies on the CU's other workgroup instead.
  constexpr bool PREF = STRIDE == 1;
  uint2 dv[ND][2], xv[NX][2][2];
  auto fetch = [&](int unit) __attribute__((always_inline)) {
    const int ch = unit % a.nchunk;
    const int nv = unit / a.nchunk;              // n * V + v
    const int t0 = ch * TC;
    const unsigned short* src = a.dyH + ((long)nv * a.M + m0) * Tp + t0;
    const unsigned short* sx = a.xH + ((long)nv * a.C + c0) * SETS * Tp + (t0 - 4);   // rows: (channel, parity)
    int tv = tid;                                // (laundered: see the bf16 kernel)
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tv + u * NT, r = i / (TC / 4), q = i - r * (TC / 4);
      const bool ok = (m0 + r < a.M) && (t0 + 4 * q < Tp);
      const int off = ok ? r * Tp + 4 * q : 0;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        dv[u][pl] = *reinterpret_cast<const uint2*>(src + pl * a.dy_plane + off);
        if (!ok) dv[u][pl] = make_uint2(0u, 0u);
      }
    }
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int i = min(tv + u * NT, NXG - 1), r = i / (XW / 8), k = i - r * (XW / 8);      // r = channel * SETS + parity
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const int t = t0 - 4 + 8 * k + 4 * hq;
        const bool ok = (c0 + r / SETS < a.C) && t >= 0 && t < Tp;
        const int off = ok ? r * Tp + 8 * k + 4 * hq : 4;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          xv[u][hq][pl] = *reinterpret_cast<const uint2*>(sx + pl * a.x_plane + off);
          if (!ok) xv[u][hq][pl] = make_uint2(0u, 0u);
        }
      }
    }
  };
  if (PREF && u_begin < u_end) fetch(u_begin);
  for (int unit = u_begin; unit < u_end; ++unit) {
    if (!PREF) fetch(unit);
    __syncthreads();                             // the previous unit's fragment reads are done
    int tv = tid;
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tv + u * NT, r = i / (TC / 4), q = i - r * (TC / 4);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<uint2*>(dyL + (pl * BM + r) * DPH + 4 * q) = dv[u][pl];
    }
#pragma unroll
    for (int u = 0; u < NX; ++u) {               // (tail threads rewrite the last group with its own value)
      const int i = min(tv + u * NT, NXG - 1), r = i / (XW / 8), k = i - r * (XW / 8);
      const int row = (r % SETS) * CBW + r / SETS;           // LDS rows: [plane][parity][channel]
#pragma unroll
      for (int hq = 0; hq < 2; ++hq)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          *reinterpret_cast<uint2*>(xP + (pl * SETS * CBW + row) * XPB + 8 * k + 4 * hq) = xv[u][hq][pl];
    }
    __syncthreads();
    if (PREF && unit + 1 < u_end) fetch(unit + 1);
#pragma unroll 1
    for (int ks = 0; ks < TC / 16; ++ks) {
      // A fragment: 8 consecutive frames of this lane's dy row ; B window: 16 consecutive frames of its x row
      const f16x8 ah = *reinterpret_cast<const f16x8*>(arow + 16 * ks);
      const f16x8 al = *reinterpret_cast<const f16x8*>(arow + BM * DPH + 16 * ks);
      u32x4 bw[SETS][2][2];                        // 16-frame window of each plane (of each parity row)
#pragma unroll
      for (int st = 0; st < SETS; ++st)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            bw[st][pl][i] = *reinterpret_cast<const u32x4*>(brow + (pl * SETS + st) * CBW * XPB + 16 * ks + 8 * i);
      unsigned p0[8], p1[8], o0[8], o1[8];         // even-row (or only) window, odd-row window
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        p0[i] = bw[0][0][i >> 2][i & 3];
        p1[i] = bw[0][1][i >> 2][i & 3];
        o0[i] = bw[SETS - 1][0][i >> 2][i & 3];
        o1[i] = bw[SETS - 1][1][i >> 2][i & 3];
      }
      if (STRIDE == 1) {                           // tap k = window shift k
        acc[0] = tap_mfma_f16<0>(ah, al, p0, p1, acc[0]);
        acc[1] = tap_mfma_f16<1>(ah, al, p0, p1, acc[1]);
        acc[2] = tap_mfma_f16<2>(ah, al, p0, p1, acc[2]);
        acc[3] = tap_mfma_f16<3>(ah, al, p0, p1, acc[3]);
        acc[4] = tap_mfma_f16<4>(ah, al, p0, p1, acc[4]);
        acc[5] = tap_mfma_f16<5>(ah, al, p0, p1, acc[5]);
        acc[6] = tap_mfma_f16<6>(ah, al, p0, p1, acc[6]);
        acc[7] = tap_mfma_f16<7>(ah, al, p0, p1, acc[7]);
        acc[8] = tap_mfma_f16<8>(ah, al, p0, p1, acc[8]);
      } else {                                     // even taps: xE shifted by 4 + (k-4)/2 ; odd taps: xO by 4 + (k-5)/2
        acc[0] = tap_mfma_f16<2>(ah, al, p0, p1, acc[0]);
        acc[1] = tap_mfma_f16<2>(ah, al, o0, o1, acc[1]);
        acc[2] = tap_mfma_f16<3>(ah, al, p0, p1, acc[2]);
        acc[3] = tap_mfma_f16<3>(ah, al, o0, o1, acc[3]);
        acc[4] = tap_mfma_f16<4>(ah, al, p0, p1, acc[4]);
        acc[5] = tap_mfma_f16<4>(ah, al, o0, o1, acc[5]);
        acc[6] = tap_mfma_f16<5>(ah, al, p0, p1, acc[6]);
        acc[7] = tap_mfma_f16<5>(ah, al, o0, o1, acc[7]);
        acc[8] = tap_mfma_f16<6>(ah, al, p0, p1, acc[8]);
      }
    }
  }

  float s_dy, inv_dy, s_x, inv_x;
  f16_range_scale(a.dy_absmax, s_dy, inv_dy);
  f16_range_scale(a.x_absmax, s_x, inv_x);
  const float inv = inv_dy * inv_x;
  float* dst = a.part + (long)split * 9 * a.M * a.C;
#pragma unroll
  for (int z = 0; z < 9; ++z)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int m = m0 + rt * 32 + mfma_row(j, h);
      const int c = c0 + ct * 32 + lr;
      if (m < a.M && c < a.C) dst[((long)z * a.M + m) * a.C + c] = acc[z][j] * inv;
    }
}

struct W9Geom {
  int Tp, nchunk, units, nmb, ncb, nsplit, units_per_split, rt;
  size_t slab_bytes, dyT_bytes, xT_bytes, smem_bytes, smem_f16;
};

inline W9Geom w9_geometry(int N, int M, int C, int V, int T, int stride) {      // T = OUTPUT frames
  W9Geom g;
  g.rt = (M % 128 == 0) ? 4 : 2;
  const int BM = g.rt * 32, CBW = (4 / g.rt) * 32;
  g.Tp = (T + 7) & ~7;                               // (the fp16 transposer writes 8 frames per lane)
  g.nchunk = (T + TC - 1) / TC;
  g.units = N * V * g.nchunk;
  g.nmb = (M + BM - 1) / BM;
  g.ncb = (C + CBW - 1) / CBW;
  const int tiles = g.nmb * g.ncb;
  int want = 512 / tiles;                          // two 4-wave workgroups per CU
  if (want < 1) want = 1;
  if (want > g.units) want = g.units;
  g.units_per_split = (g.units + want - 1) / want;
  g.nsplit = (g.units + g.units_per_split - 1) / g.units_per_split;
  g.slab_bytes = (size_t)g.nsplit * 9 * M * C * 4;
  g.dyT_bytes = (size_t)N * V * M * g.Tp * 4;
  g.xT_bytes = (size_t)N * V * C * stride * g.Tp * 4;
  g.smem_bytes = (size_t)BM * DP * 4 + (size_t)3 * stride * CBW * XPB * 2;
  g.smem_f16 = (size_t)2 * BM * DPH * 2 + (size_t)2 * stride * CBW * XPB * 2;
  return g;
}

int w9_transpose(const float* in, float* out, int N, int R, int T, int V, int Tp, int S, hipStream_t s,
                 const float* absmax = nullptr) {
  TrArgs t;
  t.in = in; t.out = out; t.N = N; t.R = R; t.T = T; t.V = V; t.Tp = Tp; t.S = S;
  t.absmax = absmax; t.plane = (long)N * V * R * S * Tp;
  if (absmax) {                                  // fp16 plane pairs (Tp is a multiple of 8)
    // 128-frame tiles where the source frames fill them (measured: T = 75 and 300 gain, T = 150 = 128 + 22 loses)
    const int rem = T % 128;
    if (T <= 128 || rem == 0 || rem > 64) {
      constexpr int RB = 4, TT = 128;
      const int ntt = (Tp * S + TT - 1) / TT, nrb = (R + RB - 1) / RB;
      const size_t smem = (size_t)RB * (TT * V + 4) * 4;
      hipLaunchKernelGGL((tv_transpose_kernel<true, RB, TT, 8>), dim3((unsigned)(N * nrb * ntt)), dim3(256), smem, s, t);
    } else {
      constexpr int RB = 8, TT = 64;
      const int ntt = (Tp * S + TT - 1) / TT, nrb = (R + RB - 1) / RB;
      const size_t smem = (size_t)RB * (TT * V + 4) * 4;
      hipLaunchKernelGGL((tv_transpose_kernel<true, RB, TT, 8>), dim3((unsigned)(N * nrb * ntt)), dim3(256), smem, s, t);
    }
  } else {
    const int ntt = (Tp * S + 63) / 64, nrb = (R + 7) / 8;
    const size_t smem = (size_t)8 * (64 * V + 4) * 4;
    hipLaunchKernelGGL((tv_transpose_kernel<false, 8, 64, 4>), dim3((unsigned)(N * nrb * ntt)), dim3(256), smem, s, t);
  }
  return agcn_check_launch();
}

template <int RT, int CT, int STRIDE>
int w9_launch_f16(const W9Args& a, const W9Geom& g, hipStream_t s) {
  constexpr auto kern = wgrad9_f16_kernel<RT, CT, STRIDE>;
  int rc = agcn_allow_big_lds<kern>();
  if (rc) return rc;
  AGCN_NOTE_KERNEL("wgrad9_f16_kernel<%d, %d, %d>", RT, CT, STRIDE);
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.nmb * g.ncb), (unsigned)g.nsplit), dim3(256), g.smem_f16, s, a);
  return agcn_check_launch();
}

template <int RT, int CT, int STRIDE>
int w9_launch(const W9Args& a, const W9Geom& g, hipStream_t s) {
  constexpr auto kern = wgrad9_bf16_kernel<RT, CT, STRIDE>;
  int rc = agcn_allow_big_lds<kern>();
  if (rc) return rc;
  AGCN_NOTE_KERNEL("wgrad9_bf16_kernel<%d, %d, %d>", RT, CT, STRIDE);
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.nmb * g.ncb), (unsigned)g.nsplit), dim3(256), g.smem_bytes, s, a);
  return agcn_check_launch();
}

}  // namespace

// stride-1 9-tap problems whose rows and channels tile by 64 (every unit_tcn of the AGCN/AAGCN stacks except the two
// stride-2 ones)
bool agcn_wgrad9_bf16_supported(int M, int C, int V, int stride) {
  if (M % 64 != 0 || C % 64 != 0 || V < 1 || V > 32 || !agcn_chained()) return false;
  return stride == 1 || (stride == 2 && M % 128 == 0);      // (stride 2 only with the 128-row tile: LDS budget)
}

// T = frames of x; the convolution output has (T - 1) / stride + 1 frames
size_t agcn_wgrad9_bf16_workspace(int N, int M, int C, int V, int T, int stride) {
  const W9Geom g = w9_geometry(N, M, C, V, (T - 1) / stride + 1, stride);
  return g.slab_bytes + g.dyT_bytes + g.xT_bytes + 512;      // (alignment of the copies + two scalars behind them)
}

// AGCN_WGRAD9_F16X3=0 keeps the 9-tap weight gradient on bf16x6 (A/B)
static inline bool w9_f16x3() {
  static const int on = getenv("AGCN_WGRAD9_F16X3") ? atoi(getenv("AGCN_WGRAD9_F16X3")) : 1;
  return on != 0;
}

// writes *nslabs slabs [9][M][C] at the start of ws (to be summed by the caller's slab reduction).
// dy_absmax / x_absmax: device scalars max |dy| / max |x| where their producers left them behind (f16x3 range scales; a
// reduction pass takes a missing one).
int agcn_wgrad9_bf16(const float* dy, const float* x, void* ws, size_t ws_bytes, int* nslabs, int N, int M, int C, int V,
                     int T, int stride, hipStream_t s, const float* dy_absmax, const float* x_absmax) {
  const int To = (T - 1) / stride + 1;
  const W9Geom g = w9_geometry(N, M, C, V, To, stride);
  if (g.slab_bytes + g.dyT_bytes + g.xT_bytes + 512 > ws_bytes) return AGCN_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)ws;
  float* dyT = (float*)(base + ((g.slab_bytes + 255) & ~(size_t)255));
  float* xT = (float*)((unsigned char*)dyT + g.dyT_bytes);
  if ((size_t)((unsigned char*)xT - base) + g.xT_bytes + 8 > ws_bytes) return AGCN_ERR_WORKSPACE;
  const bool f16 = agcn_npl() == 3 && w9_f16x3();
  if (f16) {
    unsigned* scr = reinterpret_cast<unsigned*>((unsigned char*)xT + g.xT_bytes);     // two scalars behind the copies
    if (!dy_absmax) {
      if (int rc = agcn_launch_absmax(dy, (long)N * M * To * V, scr, s)) return rc;
      dy_absmax = reinterpret_cast<const float*>(scr);
    }
    if (!x_absmax) {
      if (int rc = agcn_launch_absmax(x, (long)N * C * T * V, scr + 1, s)) return rc;
      x_absmax = reinterpret_cast<const float*>(scr + 1);
    }
  }
  int rc = w9_transpose(dy, dyT, N, M, To, V, g.Tp, 1, s, f16 ? dy_absmax : nullptr);
  if (rc) return rc;
  rc = w9_transpose(x, xT, N, C, T, V, g.Tp, stride, s, f16 ? x_absmax : nullptr);
  if (rc) return rc;
  W9Args a = {};
  a.dyT = dyT; a.xT = xT; a.part = (float*)ws;
  a.N = N; a.M = M; a.C = C; a.V = V; a.T = To; a.Tp = g.Tp;
  a.nchunk = g.nchunk; a.units = g.units; a.units_per_split = g.units_per_split; a.ncb = g.ncb;
  a.npl = agcn_npl();
  *nslabs = g.nsplit;
  if (f16) {
    a.dyH = reinterpret_cast<const unsigned short*>(dyT);
    a.xH = reinterpret_cast<const unsigned short*>(xT);
    a.dy_plane = (long)N * V * M * g.Tp;
    a.x_plane = (long)N * V * C * stride * g.Tp;
    a.dy_absmax = dy_absmax; a.x_absmax = x_absmax;
    if (stride == 2) return w9_launch_f16<4, 1, 2>(a, g, s);
    return g.rt == 4 ? w9_launch_f16<4, 1, 1>(a, g, s) : w9_launch_f16<2, 2, 1>(a, g, s);
  }
  if (stride == 2) return w9_launch<4, 1, 2>(a, g, s);
  return g.rt == 4 ? w9_launch<4, 1, 1>(a, g, s) : w9_launch<2, 2, 1>(a, g, s);
}
