// Adaptive adjacency of unit_gcn WITHOUT the theta/phi round trip through HBM (reference agcn.py:99-101, SURVEY
// Appendix A "kernel A"): one workgroup reads a frame tile of x once, forms
//   [theta_i ; phi_i] = [Wa_i ; Wb_i] . x + [ba_i ; bb_i]                      (split-bf16 "bf16x6" MFMA, fp32-equivalent)
// for whole subsets in its accumulators, parks one subset at a time in LDS as an fp32 [row][position] tile, and
//   forward : reduces  S_i[u,v] += sum_{c',t} theta_i[c',t,u] phi_i[c',t,v]   (exact-f32 MFMA, K = (c',t) pairs)
//             into the (sample, subset, tile) slab that adj_finalize_kernel sums, scales by 1/K and soft-maxes;
//   backward: recomputes the same tile and emits dtheta[c',t,u] = sum_v dS[u,v] phi[c',t,v],
//             dphi[c',t,v] = sum_u dS[u,v] theta[c',t,u] (dS carries the 1/K) as coalesced rows of dtp, plus the
//             per-row sums of the conv_a/conv_b bias gradients.
// theta/phi (6*Ci*T*V floats per sample: 368 MB per layer at batch 64) are never written to or read from HBM.
//
// Geometry: 8 waves; wave w owns positions [32w, 32w+32) of the 256-position tile (tt = 256/V whole frames);
// a workgroup owns TM 32-row tiles = NSUB whole subsets (2*Ci rows each); K runs over 16-channel chunks whose
// operand images (pre-split weights from the pack kernel, x split while staged) are double-buffered in LDS behind a
// register ring PD chunks deep: one barrier per chunk, the global loads of chunks k+2..k+PD in flight during chunk k.
// Phase 2 works on 16-channel groups (16 theta + 16 phi rows), so its LDS tile is 33 KB whatever Ci is.
#include "agcn_common.h"
#include "split_f16.h"
#include "split_bf16.h"

namespace {

constexpr int CK = 16, NW = 8, NT = 512, WLR = 264, TP = 257;
constexpr int B_BYTES = 3 * 2 * WLR * 16;

struct AfArgs {
  const float* x;              // (N, C, T, V)
  const unsigned short* wp;    // packed split weights [sblk][chunk][plane][h][ml][8]
  const float* bias;           // (6*Ci) stacked conv_a/conv_b biases
  float* spart;                // fwd: (N, 3, ntiles, V, V)
  float* tp_out;               // fwd, optional: (N, 6*Ci, T, V) theta/phi kept for a backward that does not recompute
  const float* dS;             // bwd: (N, 3, V, V)
  float* dtp;                  // bwd: (N, 6*Ci, T, V)
  float* dbpart;               // bwd: (N*ntiles, 6*Ci)
  int N, C, Ci, T, V, tt, ntiles, nchunks, nsb;
  int npl;                     // 3: six split products (bf16x6) ; 1: hi*hi only (bf16)
  int dbg;                     // profiling aid (AGCN_AF_DBG): 1 = skip phase 2, 2 = skip the phase-1 loop
  unsigned* x_absmax;          // fwd, optional: receives max |x| (bit pattern of a non-negative float; zeroed by the launcher):
                               // a by-product of the one pass that reads all of x, for the f16x3 chain that reads x next
};

struct AfPackArgs {
  const float* w;              // (6*Ci, C) row-major
  unsigned short* wp;
  int M, K, nchunks;
};

template <int BM>
__global__ void __launch_bounds__(256) af_pack_kernel(const AfPackArgs p) {
  constexpr int PER_PLANE = 2 * BM * 8;
  const int ch = blockIdx.x % p.nchunks, sb = blockIdx.x / p.nchunks;
  unsigned short* dst = p.wp + (long)blockIdx.x * 3 * PER_PLANE;
  for (int e = threadIdx.x; e < PER_PLANE / 2; e += 256) {
    const int j = (e & 3) * 2;
    const int ml = (e >> 2) % BM;
    const int h = (e >> 2) / BM;
    const int m = sb * BM + ml;
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kc = ch * CK + h * 8 + j + q;
      v[q] = (m < p.M && kc < p.K) ? p.w[(long)m * p.K + kc] : 0.f;
    }
    unsigned ph, pm, pl;
    sb_split_pair(v[0], v[1], ph, pm, pl);
    const int o = (h * BM + ml) * 8 + j;
    *reinterpret_cast<unsigned*>(dst + 0 * PER_PLANE + o) = ph;
    *reinterpret_cast<unsigned*>(dst + 1 * PER_PLANE + o) = pm;
    *reinterpret_cast<unsigned*>(dst + 2 * PER_PLANE + o) = pl;
  }
}

// MODE 0: forward (scores slab) ; MODE 1: backward (dtp rows + bias-gradient partials)
// PD: chunks whose global loads are in flight ahead of the MFMAs (the per-chunk matrix work is far shorter than an HBM
// round trip, so the ring has to be several chunks deep)
// Two workgroups per CU (<= 128 VGPRs) where accumulators + ring allow it, else one with a deeper ring.
template <int TM, int NSUB, int MODE, int PD>
__global__ void __launch_bounds__(NT, (TM * 16 + PD * 8 + 60 <= 128) ? 4 : 2) adj_fused_kernel(const AfArgs a) {
  constexpr int BM = TM * 32;
  constexpr int RT = TM / NSUB;                 // 32-row tiles per subset
  constexpr int RS = RT * 32;                   // rows per subset = 2*Ci
  constexpr int CI = RS / 2;
  constexpr int NG = CI / 16;                   // 16-channel groups of a subset (phase 2 works one group at a time)
  static_assert(TM % NSUB == 0 && CI % 16 == 0, "whole subsets per workgroup, 16-channel groups");
  constexpr int A_PLANE = 2 * BM * 16;          // bytes per plane of the A image
  constexpr int A_BYTES = 3 * A_PLANE;
  constexpr int A16 = A_BYTES / 16;
  constexpr int EA = (A16 + NT - 1) / NT;
  constexpr int BUF = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int sblk = bid % a.nsb;
  const int nt_id = bid / a.nsb;
  const int n = nt_id / a.ntiles, tile_id = nt_id - n * a.ntiles;
  const int V = a.V, tt = a.tt, t0 = tile_id * tt;
  const int tvalid = min(tt, a.T - t0);
  const int nvalid = tvalid * V;
  const long P = (long)a.T * V;
  const int g0 = t0 * V;

  f32x16 acc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;

  // ---- phase 1: [theta;phi] tile = W . x ; LDS images double-buffered, register ring PD chunks deep ----
  const int hb = wave & 1, wq = wave >> 1;      // staging role: channel half, 64-position quarter
  u32x4 ra[PD][EA];
  float rb[PD][8];
  const u32x4* wp4 = reinterpret_cast<const u32x4*>(a.wp) + (long)sblk * a.nchunks * A16;
  const int rpos = wq * 64 + lane;              // staged position of this lane
  const bool okp = rpos < nvalid;
  const float* xrow = a.x + (long)n * a.C * P + (okp ? (g0 + rpos) : 0);

  unsigned xmax = 0;                            // running max |x| of the elements this thread stages (forward, row block 0)
  const bool track_amax = MODE == 0 && a.x_absmax != nullptr && sblk == 0;
  auto issue_loads = [&](int ch, u32x4 (&qa)[EA], float (&qb)[8]) __attribute__((always_inline)) {
    const u32x4* src = wp4 + (long)ch * A16;
#pragma unroll
    for (int u = 0; u < EA; ++u) qa[u] = src[min(tid + u * NT, A16 - 1)];
    const int kc0 = ch * CK + hb * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) qb[c] = xrow[(long)min(kc0 + c, a.C - 1) * P];
  };
  auto commit_lds = [&](int ch, const u32x4 (&qa)[EA], const float (&qb)[8]) __attribute__((always_inline)) {
    unsigned char* Ab = smem + (ch & 1) * BUF;
    unsigned char* Bb = Ab + A_BYTES;
#pragma unroll
    for (int u = 0; u < EA; ++u) reinterpret_cast<u32x4*>(Ab)[min(tid + u * NT, A16 - 1)] = qa[u];   // tail lanes rewrite unit A16-1 with its own value
    const int kc0 = ch * CK + hb * 8;
    if (track_amax) {
#pragma unroll
      for (int c = 0; c < 8; ++c) xmax = max(xmax, __float_as_uint(qb[c]) & 0x7fffffffu);   // (clamped lanes repeat real elements)
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x0 = (okp && (kc0 + 2 * c) < a.C) ? qb[2 * c] : 0.f;
      const float x1 = (okp && (kc0 + 2 * c + 1) < a.C) ? qb[2 * c + 1] : 0.f;
      unsigned a0, a1, a2;
      sb_split_pair(x0, x1, a0, a1, a2);
      ph[c] = a0; pm[c] = a1; pl[c] = a2;
    }
    *reinterpret_cast<u32x4*>(Bb + ((0 * 2 + hb) * WLR + rpos) * 16) = ph;
    *reinterpret_cast<u32x4*>(Bb + ((1 * 2 + hb) * WLR + rpos) * 16) = pm;
    *reinterpret_cast<u32x4*>(Bb + ((2 * 2 + hb) * WLR + rpos) * 16) = pl;
  };

  // Straight-line ring (nchunks % PD == 0, host-checked): every load/commit below is unconditional (chunk indices are
  // clamped, so the tail re-loads the last chunk and re-writes an LDS buffer nobody reads any more) -- the number of
  // loads in flight is then the same on every path and the compiler's vmcnt waits leave PD-1 chunks outstanding.
  const int nchunks = a.nchunks, lastc = nchunks - 1;
#pragma unroll
  for (int d = 0; d < PD; ++d) issue_loads(min(d, lastc), ra[d], rb[d]);
  commit_lds(0, ra[0], rb[0]);
  issue_loads(min(PD, lastc), ra[0], rb[0]);
  __syncthreads();
  const int bq = wave * 32 + lr;                // this lane's B column (position)
  for (int ch0 = 0; ch0 < ((a.dbg & 2) ? 0 : nchunks); ch0 += PD) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int ch = ch0 + d;                   // chunk c sits in ring slot c % PD
      commit_lds(ch + 1, ra[(d + 1) % PD], rb[(d + 1) % PD]);     // other LDS buffer: free since the last barrier
      issue_loads(min(ch + 1 + PD, lastc), ra[(d + 1) % PD], rb[(d + 1) % PD]);
      const unsigned char* Ab = smem + (ch & 1) * BUF;
      const unsigned char* Bb = Ab + A_BYTES;
      bf16x8 bf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bf16x8*>(Bb + ((pl * 2 + h) * WLR + bq) * 16);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        bf16x8 af[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          af[pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_PLANE + ((h * BM) + tm * 32 + lr) * 16);
        acc[tm] = sb_mfma6(af[0], af[1], af[2], bf[0], bf[1], bf[2], acc[tm], a.npl);
      }
      __syncthreads();
    }
  }

  if (track_amax) {
    // max is order-independent (deterministic).  One atomic per WAVE at most, and none once the running maximum in memory
    // has passed the wave's own (thousands of atomics on one address would serialise in L2)
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) xmax = max(xmax, (unsigned)__shfl_xor((int)xmax, k));
    if (lane == 0 && xmax > __hip_atomic_load(a.x_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.x_absmax, xmax);
  }
  // bias of the rows this lane holds
  if (a.bias) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[tm][j] += a.bias[sblk * BM + tm * 32 + mfma_row(j, h)];
  }

  if (a.dbg & 1) {                              // profiling aid: keep the accumulators live, skip phase 2
    float sum = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += acc[tm][j];
    if (sum == 1.2345e-30f) a.spart[0] = sum;
    return;
  }
  // ---- phase 2: one 16-channel group (16 theta rows + 16 phi rows) at a time through the fp32 LDS tile [32][TP],
  //      which overlays the staging buffers ----
  float* tile = reinterpret_cast<float*>(smem);
  float* aux = tile + 32 * TP;                  // fwd: red[NW][V*V] ; bwd: B1, B2, row sums
  const int VV = V * V;
  const int lc = min(lr, V - 1);
  const int VS = (V + 1) >> 1, VP = 2 * VS;

#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const int isub = sblk * NSUB + s;           // global subset index
    f32x16 d;                                   // fwd: this wave's partial of S_isub
#pragma unroll
    for (int j = 0; j < 16; ++j) d[j] = 0.f;
    if (MODE == 1) {
      __syncthreads();                          // previous readers of aux are done
      float* B1 = aux;                          // [VP][32]: B1[k=v][col=u] = dS[u][v]
      float* B2 = aux + VP * 32;                // [VP][32]: B2[k=u][col=v] = dS[u][v]
      const float* dsn = a.dS + ((long)n * 3 + isub) * VV;
      for (int e = tid; e < VP * 32; e += NT) {
        const int k = e >> 5, col = e & 31;
        const bool ok = k < V && col < V;
        B1[e] = ok ? dsn[col * V + k] : 0.f;
        B2[e] = ok ? dsn[k * V + col] : 0.f;
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // theta channels 16g..16g+15 = subset rows 16g.., phi = subset rows CI+16g..: (row tile, register half) of each
      const int th_tile = s * RT + (16 * g) / 32, th_half = ((16 * g) / 16) & 1;
      const int ph_tile = s * RT + (CI + 16 * g) / 32, ph_half = ((CI + 16 * g) / 16) & 1;
      if (s > 0 || g > 0 || MODE == 0) __syncthreads();   // previous group's readers are done with the tile
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r16 = (e & 3) + 8 * (e >> 2) + 4 * h;   // row within the 16-row half
        tile[r16 * TP + bq] = (bq < nvalid) ? acc[th_tile][8 * th_half + e] : 0.f;
        tile[(16 + r16) * TP + bq] = (bq < nvalid) ? acc[ph_tile][8 * ph_half + e] : 0.f;
      }
      __syncthreads();

      if (MODE == 0 && a.tp_out) {              // training forward: leave theta/phi in HBM as coalesced rows
        for (int r = wave; r < 32; r += NW) {
          const int srow = (r < 16) ? (16 * g + r) : (CI + 16 * g + r - 16);      // row within the subset
          float* drow = a.tp_out + ((long)n * 3 * RS + (long)isub * RS + srow) * P + g0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = lane + 64 * u;
            if (q < nvalid) drow[q] = tile[r * TP + q];
          }
        }
      }
      if (MODE == 0) {
        // channels c = wave, wave + 8 of the group ; k pairs = frames (2f, 2f+1) of one channel
        const int npf = (tt + 1) >> 1;
        for (int c = wave; c < 16; c += NW) {
          const float* th = tile + c * TP + lc;
          const float* ph = tile + (16 + c) * TP + lc;
          for (int f = 0; f < npf; ++f) {
            const int t = 2 * f + h;
            const bool ok = lr < V && t < tt;
            const int o = ok ? t * V : 0;
            float av = th[o], bv = ph[o];
            av = ok ? av : 0.f;
            bv = ok ? bv : 0.f;
            d = mfma32(av, bv, d);
          }
        }
      } else {
        const float* B1 = aux;
        const float* B2 = aux + VP * 32;
        float* rows = aux + 2 * VP * 32;        // [32] row sums
        const int nrows = 16 * tt;              // (c', t) rows of the group's theta (and phi)
        const int nrt = (nrows + 31) >> 5;      // <= 8 (tt <= 16): two result tiles per wave at most
        // result tiles 0..nrt-1: dtheta (operand rows: phi, B1) ; nrt..2nrt-1: dphi (operand rows: theta, B2)
        f32x16 dd[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int tl = wave + NW * k;
#pragma unroll
          for (int j = 0; j < 16; ++j) dd[k][j] = 0.f;
          if (tl < 2 * nrt) {
            const int which = tl / nrt, rt = tl - which * nrt;
            const float* Bf = which == 0 ? B1 : B2;
            const int row = min(rt * 32 + lr, nrows - 1);
            const int c = row / tt, t = row - c * tt;
            const float* src = tile + ((which == 0 ? 16 : 0) + c) * TP + t * V;
            for (int s2 = 0; s2 < VS; ++s2) {
              const int k2 = 2 * s2 + h;
              float av = src[min(k2, V - 1)];
              av = (k2 < V) ? av : 0.f;
              dd[k] = mfma32(av, Bf[k2 * 32 + lr], dd[k]);
            }
          }
        }
        __syncthreads();                        // every wave is done reading the operand rows
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int tl = wave + NW * k;
          if (tl < 2 * nrt) {
            const int which = tl / nrt, rt = tl - which * nrt;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int r2 = rt * 32 + mfma_row(j, h);
              const int c = r2 / tt, t = r2 - c * tt;
              if (r2 < nrows && lr < V) tile[(which * 16 + c) * TP + t * V + lr] = dd[k][j];
            }
          }
        }
        __syncthreads();
        // coalesced rows of dtp + per-row sums for the bias gradients
        for (int r = wave; r < 32; r += NW) {
          const int srow = (r < 16) ? (16 * g + r) : (CI + 16 * g + r - 16);      // row within the subset
          float* drow = a.dtp + ((long)n * 3 * RS + (long)isub * RS + srow) * P + g0;
          float sum = 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = lane + 64 * u;
            if (q < nvalid) {
              const float val = tile[r * TP + q];
              drow[q] = val;
              sum += val;
            }
          }
          sum = half_sum(sum);
          sum += __shfl_xor(sum, 32);
          if (lane == 0 && a.dbpart)
            a.dbpart[((long)n * a.ntiles + tile_id) * 3 * RS + (long)isub * RS + srow] = sum;
        }
        (void)rows;
      }
    }
    if (MODE == 0) {
      float* red = aux;
      __syncthreads();                          // (aux is free: nothing else uses it in the forward)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int u = mfma_row(j, h);
        if (u < V && lr < V) red[wave * VV + u * V + lr] = d[j];
      }
      __syncthreads();
      float* dst = a.spart + (((long)n * 3 + isub) * a.ntiles + tile_id) * VV;
      for (int e = tid; e < VV; e += NT) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[w * VV + e];
        dst[e] = sum;
      }
    }
  }
}

struct AfGeom {
  int tt, ntiles, nchunks, nsb;
  size_t smem_bytes, pack_bytes;
};

inline AfGeom af_geometry(int C, int T, int V, int BM, int nsub) {
  AfGeom g;
  g.tt = 256 / V;
  if (g.tt > T) g.tt = T;
  g.ntiles = (T + g.tt - 1) / g.tt;
  g.nchunks = (C + CK - 1) / CK;
  g.nsb = 3 / nsub;
  const size_t a_bytes = (size_t)3 * 2 * BM * 16;
  const size_t stage = 2 * (a_bytes + B_BYTES);
  const int VP = 2 * ((V + 1) / 2);
  const size_t aux_f = (size_t)NW * V * V, aux_b = (size_t)2 * VP * 32 + 32;
  const size_t ph2 = ((size_t)32 * TP + (aux_f > aux_b ? aux_f : aux_b)) * 4;
  g.smem_bytes = ((stage > ph2 ? stage : ph2) + 15) & ~(size_t)15;
  g.pack_bytes = (size_t)g.nsb * g.nchunks * a_bytes;
  return g;
}

// (TM, NSUB) for a given Ci: whole subsets per workgroup, <= 128 VGPRs so that two workgroups share a CU
inline bool af_shape(int Ci, int& tm, int& nsub) {
  if (Ci == 16) { tm = 3; nsub = 3; return true; }
  if (Ci == 32) { tm = 2; nsub = 1; return true; }
  if (Ci == 64) { tm = 4; nsub = 1; return true; }
  return false;
}

template <int TM, int NSUB, int MODE, int PD>
int af_launch(AfArgs a, const AfGeom& g, hipStream_t s) {
  constexpr auto kern = adj_fused_kernel<TM, NSUB, MODE, PD>;
  int rc = agcn_allow_big_lds<kern>();
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * g.ntiles * g.nsb)), dim3(NT), g.smem_bytes, s, a);
  return agcn_check_launch();
}

template <int TM, int NSUB>
int af_pack(const float* wab, void* ws, int M, int K, const AfGeom& g, hipStream_t s) {
  AfPackArgs pk;
  pk.w = wab; pk.wp = (unsigned short*)ws; pk.M = M; pk.K = K; pk.nchunks = g.nchunks;
  hipLaunchKernelGGL((af_pack_kernel<TM * 32>), dim3(g.nsb * g.nchunks), dim3(256), 0, s, pk);
  return agcn_check_launch();
}

template <int TM, int NSUB>
int af_go(int mode, const AfArgs& a, const float* wab, void* ws, const AfGeom& g, hipStream_t s) {
  int rc = af_pack<TM, NSUB>(wab, ws, 6 * a.Ci, a.C, g, s);
  if (rc) return rc;
  // ring depth: the deepest of {4, 2, 1} that divides the chunk count (4 only where two workgroups still fit a CU or
  // the K loop is long enough to need it)
  const int pd = (g.nchunks % 4 == 0 && (TM <= 2 || g.nchunks >= 8)) ? 4 : (g.nchunks % 2 == 0 ? 2 : 1);
  if (pd == 4) return mode == 0 ? af_launch<TM, NSUB, 0, 4>(a, g, s) : af_launch<TM, NSUB, 1, 4>(a, g, s);
  if (pd == 2) return mode == 0 ? af_launch<TM, NSUB, 0, 2>(a, g, s) : af_launch<TM, NSUB, 1, 2>(a, g, s);
  return mode == 0 ? af_launch<TM, NSUB, 0, 1>(a, g, s) : af_launch<TM, NSUB, 1, 1>(a, g, s);
}

int af_run(int mode, AfArgs a, const float* wab, void* ws, size_t ws_bytes, hipStream_t s) {
  int tm, nsub;
  if (!af_shape(a.Ci, tm, nsub)) return AGCN_ERR_UNSUPPORTED;
  const AfGeom g = af_geometry(a.C, a.T, a.V, tm * 32, nsub);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if (g.pack_bytes > ws_bytes) return AGCN_ERR_WORKSPACE;
  a.tt = g.tt; a.ntiles = g.ntiles; a.nchunks = g.nchunks; a.nsb = g.nsb;
  a.wp = (const unsigned short*)ws;
  { const char* e = getenv("AGCN_AF_DBG"); a.dbg = e ? atoi(e) : 0; }
  a.npl = agcn_npl();
  if (tm == 3) return af_go<3, 3>(mode, a, wab, ws, g, s);
  if (tm == 2) return af_go<2, 1>(mode, a, wab, ws, g, s);
  return af_go<4, 1>(mode, a, wab, ws, g, s);
}

bool af_supported(int C, int Ci, int T, int V) {
  int tm, nsub;
  if (V < 16 || V > 32 || C < 1 || T < 1 || !af_shape(Ci, tm, nsub)) return false;   // tt = 256/V <= 16 frames
  if (!agcn_chained()) return false;            // AGCN_GEMM=f32 / bf16x3 keep the two-kernel path
  return af_geometry(C, T, V, tm * 32, nsub).smem_bytes <= 160 * 1024;
}

}  // namespace

extern "C" {

int agcn_adjacency_fused_supported(int C, int Ci, int T, int V) { return af_supported(C, Ci, T, V) ? 1 : 0; }

size_t agcn_adjacency_fused_workspace(int C, int Ci) {
  int tm, nsub;
  if (!af_shape(Ci, tm, nsub)) return 256;
  const size_t own = af_geometry(C, 1, 25, tm * 32, nsub).pack_bytes + 256, ws = agcn_adj_ws_workspace(C, Ci) + 256;
  return own > ws ? own : ws;
}

int agcn_adjacency_fused_fwd_ex(const float* x, const float* wab, const float* bab, const float* A, const float* PA,
                                const float* alpha, float* tp_out, float* spart, float* P, float* adj, float* x_absmax_out,
                                const float* x_absmax_in, void* workspace, size_t workspace_bytes, int N, int C, int Ci, int T,
                                int V, void* stream);
int agcn_adjacency_fused_fwd(const float* x, const float* wab, const float* bab, const float* A, const float* PA,
                             const float* alpha, float* tp_out, float* spart, float* P, float* adj, void* workspace,
                             size_t workspace_bytes, int N, int C, int Ci, int T, int V, void* stream) {
  return agcn_adjacency_fused_fwd_ex(x, wab, bab, A, PA, alpha, tp_out, spart, P, adj, nullptr, nullptr, workspace,
                                     workspace_bytes, N, C, Ci, T, V, stream);
}
// x_absmax_out (optional, 4 bytes): receives max |x|, a by-product of the pass (for the f16x3 chain that reads x next).
// x_absmax_in (optional): max |x| when the producer of x already took it.  The layers whose stacked weights fit in LDS
// (adj_ws.hip) run on the persistent f16x3 kernel, which needs the maximum up front: taken by a reduction pass when absent.
int agcn_adjacency_fused_fwd_ex(const float* x, const float* wab, const float* bab, const float* A, const float* PA,
                                const float* alpha, float* tp_out, float* spart, float* P, float* adj, float* x_absmax_out,
                                const float* x_absmax_in, void* workspace, size_t workspace_bytes, int N, int C, int Ci, int T,
                                int V, void* stream) {
  if (!x || !wab || !PA || !spart || !P || !adj || !workspace || N <= 0 || C <= 0 || Ci <= 0 || T <= 0 || V <= 0 ||
      V > 32)
    return AGCN_ERR_ARG;
  if (!af_supported(C, Ci, T, V)) return AGCN_ERR_UNSUPPORTED;
  if (agcn_adj_ws_supported(N, C, Ci, T, V)) {
    const size_t img = agcn_adj_ws_workspace(C, Ci);
    if (workspace_bytes < img + 16) return AGCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const float* amax = x_absmax_in;
    if (!amax) {
      float* slot = x_absmax_out ? x_absmax_out : reinterpret_cast<float*>(static_cast<char*>(workspace) + img);
      if (int rc = agcn_launch_absmax(x, (long)N * C * T * V, reinterpret_cast<unsigned*>(slot), s)) return rc;
      amax = slot;
    } else if (x_absmax_out && x_absmax_out != x_absmax_in) {
      if (hipMemcpyAsync(x_absmax_out, x_absmax_in, 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return AGCN_ERR_ARG;
    }
    const int tt = 256 / V > T ? T : 256 / V, slots = (T + tt - 1) / tt;
    int nused = 0;
    if (int rc = agcn_adj_ws_scores(x, wab, bab, tp_out, spart, slots, &nused, amax, workspace, img, N, C, Ci, T, V, s))
      return rc;
    return agcn_adj_finalize(spart, A, PA, alpha, P, adj, N, Ci, T, V, s, nused);
  }
  if (x_absmax_out && hipMemsetAsync(x_absmax_out, 0, 4, (hipStream_t)stream) != hipSuccess) return AGCN_ERR_ARG;
  AfArgs a = {};
  a.x_absmax = reinterpret_cast<unsigned*>(x_absmax_out);
  a.x = x; a.bias = bab; a.spart = spart; a.tp_out = tp_out;
  a.N = N; a.C = C; a.Ci = Ci; a.T = T; a.V = V;
  int rc = af_run(0, a, wab, workspace, workspace_bytes, (hipStream_t)stream);
  if (rc) return rc;
  return agcn_adj_finalize(spart, A, PA, alpha, P, adj, N, Ci, T, V, (hipStream_t)stream);
}

int agcn_adjacency_fused_bwd_scores(const float* x, const float* wab, const float* bab, const float* dS, float* dtp,
                                    float* dbpart, void* scratch, float* db, void* workspace, size_t workspace_bytes,
                                    int N, int C, int Ci, int T, int V, void* stream) {
  if (!x || !wab || !dS || !dtp || !dbpart || !scratch || !db || !workspace || N <= 0 || C <= 0 || Ci <= 0 ||
      T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  if (!af_supported(C, Ci, T, V)) return AGCN_ERR_UNSUPPORTED;
  AfArgs a = {};
  a.x = x; a.bias = bab; a.dS = dS; a.dtp = dtp; a.dbpart = dbpart;
  a.N = N; a.C = C; a.Ci = Ci; a.T = T; a.V = V;
  int rc = af_run(1, a, wab, workspace, workspace_bytes, (hipStream_t)stream);
  if (rc) return rc;
  const int tt = 256 / V > T ? T : 256 / V;
  return agcn_colsum(dbpart, N * ((T + tt - 1) / tt), 6 * Ci, scratch, db, stream);
}

}  // extern "C"
