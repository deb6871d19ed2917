// Adaptive adjacency of unit_gcn (reference agcn.py:99-101: theta = conv_a(x), phi = conv_b(x), S = theta^T phi / K), forward,
// as a PERSISTENT weight-stationary kernel for the layers whose stacked conv_a/conv_b weights fit in LDS (Ci <= 32, C <= 128:
// l2..l7 of the AGCN stack).  Round 2's tile-per-workgroup kernel (adj_fused.hip, kept for the wider layers and for the
// recomputing backward) spends most of a 256-position workgroup's life on its prologue (weight chunks through a register
// ring), on one barrier per 16-channel chunk and on a per-tile slab that a finalize pass sums again.  Here one workgroup per
// CU (8 waves; wave w owns 32 of the tile's 256 positions) walks the frames [f0, f1) of one sample:
//   phase 1  [theta_i ; phi_i] = [Wa_i ; Wb_i] . x  for all three subsets, f16x3 arithmetic (two fp16 planes per operand,
//            three products, x range-scaled by the tensor maximum): the weight image (6*Ci x C, 24..96 KB) stays in LDS for
//            the workgroup's life; the x fragments (lane = position, 8 consecutive channels) come STRAIGHT from global
//            memory -- 64 lanes = two coalesced 128-byte row segments per load -- into ONE register set that is reloaded
//            with the next tile's values as soon as a k-step has consumed it (in flight during the rest of the tile);
//   phase 2  the accumulators (+ bias) pass through an fp32 LDS tile [2*Ci rows][257] per subset (or all three at once when
//            they fit); from there the optional theta/phi copy for the backward leaves as whole rows, and
//            S_i[u,v] += sum_{c',t} theta[c',t,u] phi[c',t,v] runs on the exact-f32 MFMA into accumulators that stay in
//            REGISTERS across tiles: one slab per (sample, frame split) instead of one per tile.
#include "agcn_common.h"
#include "split_f16.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int AW_NW = 8, AW_NT = AW_NW * 64;

struct AwArgs {
  const float* x;              // (N, C, T, V)
  const unsigned short* wp;    // packed fp16 planes [ks][plane][h][m][8]: k = 16 ks + 8 h + e
  const float* bias;           // (6*Ci) or null
  float* spart;                // (N, 3, slots, V, V): this kernel writes slots [0, nsplit)
  float* tp_out;               // optional (N, 6*Ci, T, V)
  const float* x_absmax;       // device scalar max |x|
  int N, C, Ci, T, V;
  int FT, nsplit, fper, slots;
};

struct AwPackArgs {
  const float* w;              // (6*Ci, C)
  unsigned short* wp;
  int M, K, nks;
};

// one thread per (ks, h, m) 8-element run
__global__ void __launch_bounds__(256) aw_pack_kernel(const AwPackArgs p) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= p.nks * 2 * p.M) return;
  const int m = e % p.M, h = (e / p.M) & 1, ks = e / (2 * p.M);
  u32x4 ph, pl;
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2) {
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 16 * ks + 8 * h + 2 * e2 + q;
      v[q] = k < p.K ? p.w[(long)m * p.K + k] : 0.f;
    }
    unsigned a, b;
    split_pair_f16(v[0] * F16_W_SCALE, v[1] * F16_W_SCALE, a, b);
    ph[e2] = a; pl[e2] = b;
  }
  u32x4* dst = reinterpret_cast<u32x4*>(p.wp);
  dst[((long)(ks * 2 + 0) * 2 + h) * p.M + m] = ph;
  dst[((long)(ks * 2 + 1) * 2 + h) * p.M + m] = pl;
}

// TMS: 32-row tiles per subset (2*Ci / 32: 1 for Ci = 16, 2 for Ci = 32); KS: 16-channel k-steps (C / 16);
// NSR: subsets per phase-2 round (3 when the LDS tile holds all of them, else 1)
// TP: row stride of the fp32 tile (>= FT * V: 252 serves V = 25 and V = 18; a constant, so that the tile's rows are
// immediate offsets from one address register)
template <int TMS, int KS, int NSR, int TP>
__global__ void __launch_bounds__(AW_NT, 2) adj_ws_kernel(const AwArgs a) {
  constexpr int TM = 3 * TMS;                  // accumulator tiles of a wave
  constexpr int RS = 32 * TMS, CI = RS / 2;    // rows per subset, channels of theta (and of phi)
  constexpr int MR = 3 * RS;                   // weight rows
  constexpr int W_BYTES = KS * 2 * 2 * MR * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wimg = smem;
  float* bl = reinterpret_cast<float*>(smem + W_BYTES);       // [MR] bias
  float* T2 = bl + MR;                                        // [NSR * RS][TP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int V = a.V, T = a.T;
  const long P = (long)T * V;
  const int split = blockIdx.x % a.nsplit, n = blockIdx.x / a.nsplit;
  const int f0 = split * a.fper, f1 = min(T, f0 + a.fper);
  const int FT = a.FT;

  float rs_s, rs_inv;
  f16_range_scale(a.x_absmax, rs_s, rs_inv);
  rs_inv *= F16_W_INV;                         // (the packed weights carry F16_W_SCALE)

  // ---- resident weight image ----
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wp);
    u32x4* dst = reinterpret_cast<u32x4*>(wimg);
    for (int e = tid; e < W_BYTES / 16; e += AW_NT) dst[e] = src[e];
    for (int e = tid; e < MR; e += AW_NT) bl[e] = a.bias ? a.bias[e] : 0.f;
  }

  // ---- x fragments: lane = position 32 w + lr of the tile, k-step ks: channels 16 ks + 8 h + [0, 8) ----
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long)n * a.C * P), 0, (int)((long)a.C * P * 4), 0x00020000);
  // voffset: the lane's part (half h takes the upper 8 channels of a k-step; the position is clamped into the sample so that
  // every address is inside x: the columns past a sample's end are never used); soffset: the uniform channel part
  const int xv8 = 8 * h * (int)P, P4 = (int)P * 4;
  auto load_x = [&](int t0, int ks, float (&xc)[8]) __attribute__((always_inline)) {
    const int voff = (xv8 + min(t0 * V + 32 * wave + lr, (int)P - 1)) * 4;
    int so = 16 * ks * P4;                             // (a running scalar: 8 * KS distinct products would be kept live)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("" : "+s"(so));
      xc[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, so, 0));
      so += P4;
    }
  };
  auto mfma3 = [&](const bf16x8 (&x)[2], const bf16x8 (&y)[2], f32x16 c) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[1]), __builtin_bit_cast(f16x8, y[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[0]), c, 0, 0, 0);
    return c;
  };

  // ring of RG k-steps: k-step ks lives in slot ks % RG and, once split, the slot takes k-step ks + RG (of the next tile
  // when past this one's end): RG k-steps = 3 * TM * RG matrix ops of load latency cover
  constexpr int RG = KS < 4 ? KS : 4;
  static_assert(KS % RG == 0, "ring slots are static");
  float xr[RG][8];
#pragma unroll
  for (int ks = 0; ks < RG; ++ks) load_x(f0, ks, xr[ks]);
  f32x16 d[3];                                  // S_i partial of this wave, all tiles
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) d[i][j] = 0.f;
  __syncthreads();

  for (int t0 = f0; t0 < f1; t0 += FT) {
    const int tvalid = min(FT, f1 - t0), nvalid = tvalid * V;
    const bool has_next = t0 + FT < f1;
    const int pcol = 32 * wave + lr;                   // this lane's column (position) of the tile
    // ---- phase 1: all three subsets ----
    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
      float (&xc)[8] = xr[ks % RG];
      bf16x8 bf[2];
      {
        u32x4 w0, w1;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          unsigned p0, p1;
          split_pair_f16_mix(xc[2 * e2] * rs_s, xc[2 * e2 + 1] * rs_s, p0, p1);
          w0[e2] = p0; w1[e2] = p1;
        }
        bf[0] = __builtin_bit_cast(bf16x8, w0);
        bf[1] = __builtin_bit_cast(bf16x8, w1);
      }
      if (ks + RG < KS) load_x(t0, ks + RG, xc);
      else if (has_next) load_x(t0 + FT, ks + RG - KS, xc);
      const unsigned char* wb = wimg + (long)ks * 2 * 2 * MR * 16 + (h * MR + lr) * 16;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        bf16x8 af[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(wb + (pl * 2 * MR + tm * 32) * 16);
        acc[tm] = mfma3(af, bf, acc[tm]);
      }
    }
#pragma unroll
    for (int i0 = 0; i0 < 3; i0 += NSR) {
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 2: through the fp32 tile ----
      if (t0 > f0 || i0 > 0) __syncthreads();          // the previous round's readers are done with T2
#pragma unroll
      for (int s = 0; s < NSR; ++s)
#pragma unroll
        for (int r = 0; r < TMS; ++r)
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) {
            const int row = r * 32 + 8 * j4 + 4 * h;   // rows (within the subset) of registers 4 j4 .. 4 j4 + 3
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + (i0 + s) * RS + row);
            if (pcol < TP) {
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
                T2[(s * RS + row + jj) * TP + pcol] = acc[(i0 + s) * TMS + r][4 * j4 + jj] * rs_inv + b4[jj];
            }
          }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < NSR; ++s) {
        const int i = i0 + s;
        const float* Ts = T2 + s * RS * TP;
        if (a.tp_out) {                                // theta/phi rows for the backward: whole contiguous runs
          for (int r = wave; r < RS; r += AW_NW) {
            float* drow = a.tp_out + ((long)n * 3 * RS + (long)i * RS + r) * P + (long)t0 * V;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int q = lane + 64 * u;
              if (q < nvalid) drow[q] = Ts[r * TP + q];
            }
          }
        }
        // S_i += theta^T phi: this wave's channels c = wave, wave + 8, ...; k-pairs = frames (2 f, 2 f + 1) of one channel
        const int lc = min(lr, V - 1);
        const int npf = (tvalid + 1) >> 1;
        for (int c = wave; c < CI; c += AW_NW) {
          const float* th = Ts + c * TP + lc;
          const float* ph = Ts + (CI + c) * TP + lc;
          for (int f = 0; f < npf; ++f) {
            const int t = 2 * f + h;
            const bool ok = lr < V && t < tvalid;
            const int o = ok ? t * V : 0;
            float av = th[o], bv = ph[o];
            av = ok ? av : 0.f;
            bv = ok ? bv : 0.f;
            d[i] = mfma32(av, bv, d[i]);
          }
        }
      }
    }
  }
  // ---- the eight waves' partials -> one (V x V) slab per subset ----
  __syncthreads();
  float* red = T2;                                     // [8][V*V]
  const int VV = V * V;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int u = mfma_row(j, h);
      if (u < V && lr < V) red[wave * VV + u * V + lr] = d[i][j];
    }
    __syncthreads();
    float* dst = a.spart + (((long)n * 3 + i) * a.slots + split) * VV;
    for (int e = tid; e < VV; e += AW_NT) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < AW_NW; ++w) sum += red[w * VV + e];
      dst[e] = sum;
    }
    __syncthreads();
  }
}

struct AwGeom {
  int FT, nsplit, fper, tms, ks, nsr, TP;
  size_t smem_bytes, pack_bytes;
};

inline bool aw_geometry(int N, int C, int Ci, int T, int V, AwGeom& g) {
  if (Ci != 16 && Ci != 32) return false;
  if (C != 64 && C != 128) return false;
  if (Ci == 16 && C != 64) return false;
  if (V < 8 || V > 32 || T < 1 || (long)C * T * V * 4 >= (1L << 31)) return false;
  g.tms = 2 * Ci / 32;
  g.ks = C / 16;
  g.FT = 256 / V;
  if (g.FT > T) g.FT = T;
  const int ntile = (T + g.FT - 1) / g.FT;
  long want = (256 + N - 1) / N;
  if (const char* e = getenv("AGCN_AW_SPLIT")) want = atoi(e);     // test knob (read per call)
  if (want < 1) want = 1;
  if (want > ntile) want = ntile;
  const int tps = (int)((ntile + want - 1) / want);
  g.fper = tps * g.FT;
  g.nsplit = (ntile + tps - 1) / tps;
  g.TP = g.FT * V <= 252 ? 252 : 256;
  const size_t w_bytes = (size_t)g.ks * 2 * 2 * (6 * Ci) * 16;
  g.pack_bytes = w_bytes;
  const size_t fixed = w_bytes + (size_t)6 * Ci * 4;            // weight image + bias
  const size_t t3 = (size_t)3 * 2 * Ci * g.TP * 4, t1 = (size_t)2 * Ci * g.TP * 4;
  const size_t red = (size_t)AW_NW * V * V * 4;
  g.nsr = (fixed + t3 <= 160 * 1024) ? 3 : 1;
  size_t tile = g.nsr == 3 ? t3 : t1;
  if (red > tile) tile = red;
  g.smem_bytes = fixed + tile;
  return g.smem_bytes <= 160 * 1024;
}

template <int TMS, int KS, int NSR, int TP>
int aw_launch_tp(const AwArgs& a, const AwGeom& g, hipStream_t s) {
  auto kern = adj_ws_kernel<TMS, KS, NSR, TP>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * g.nsplit)), dim3(AW_NT), g.smem_bytes, s, a);
  AGCN_NOTE_KERNEL("adj_ws_kernel<%d, %d, %d, %d>", TMS, KS, NSR, TP);
  return agcn_check_launch();
}
template <int TMS, int KS, int NSR>
int aw_launch(const AwArgs& a, const AwGeom& g, hipStream_t s) {
  return g.TP == 252 ? aw_launch_tp<TMS, KS, NSR, 252>(a, g, s) : aw_launch_tp<TMS, KS, NSR, 256>(a, g, s);
}

}  // namespace

static inline bool aw_enabled() {
  static const int on = getenv("AGCN_ADJ_WS") ? atoi(getenv("AGCN_ADJ_WS")) : 1;
  return on != 0 && agcn_chain_f16x3();       // (the fp32-equivalent default mode; AGCN_GEMM=f32 / bf16 keep the older kernels)
}

// shapes the persistent forward takes
bool agcn_adj_ws_supported(int N, int C, int Ci, int T, int V) {
  AwGeom g;
  return aw_enabled() && aw_geometry(N, C, Ci, T, V, g);
}

size_t agcn_adj_ws_workspace(int C, int Ci) { return (size_t)((C + 15) / 16) * 2 * 2 * (6 * Ci) * 16; }

// slabs only: spart (N, 3, slots, V, V) gets *nslots_used partial score matrices per (sample, subset); the caller finalises
// (agcn_adj_finalize_slots).  x_absmax: device scalar max |x| (required).
int agcn_adj_ws_scores(const float* x, const float* wab, const float* bab, float* tp_out, float* spart, int slots,
                       int* nslots_used, const float* x_absmax, void* ws, size_t ws_bytes, int N, int C, int Ci, int T, int V,
                       hipStream_t s) {
  AwGeom g;
  if (!aw_geometry(N, C, Ci, T, V, g)) return AGCN_ERR_UNSUPPORTED;
  if (g.nsplit > slots) return AGCN_ERR_UNSUPPORTED;
  if (g.pack_bytes > ws_bytes) return AGCN_ERR_WORKSPACE;
  AwPackArgs pk;
  pk.w = wab; pk.wp = (unsigned short*)ws; pk.M = 6 * Ci; pk.K = C; pk.nks = g.ks;
  const int items = g.ks * 2 * 6 * Ci;
  hipLaunchKernelGGL(aw_pack_kernel, dim3((items + 255) / 256), dim3(256), 0, s, pk);
  int rc = agcn_check_launch();
  if (rc) return rc;
  AwArgs a = {};
  a.x = x; a.wp = (const unsigned short*)ws; a.bias = bab; a.spart = spart; a.tp_out = tp_out; a.x_absmax = x_absmax;
  a.N = N; a.C = C; a.Ci = Ci; a.T = T; a.V = V;
  a.FT = g.FT; a.nsplit = g.nsplit; a.fper = g.fper; a.slots = slots;
  *nslots_used = g.nsplit;
  if (g.tms == 1 && g.ks == 4) return g.nsr == 3 ? aw_launch<1, 4, 3>(a, g, s) : aw_launch<1, 4, 1>(a, g, s);
  if (g.tms == 2 && g.ks == 4) return aw_launch<2, 4, 1>(a, g, s);
  if (g.tms == 2 && g.ks == 8) return aw_launch<2, 8, 1>(a, g, s);
  return AGCN_ERR_UNSUPPORTED;
}
