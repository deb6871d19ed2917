// Adaptive adjacency of unit_gcn (reference agcn.py:95,99-102):
//   S_i[u,v]  = (1/K) sum_{c',t} theta_i[c',t,u] * phi_i[c',t,v],  K = Ci*T
//   P_i[:,v]  = softmax_u(S_i[:,v])               (normalised over the FIRST joint index)
//   A^_i      = P_i + A_i + PA_i                   (AGCN)     |  PA_i + alpha*P_i  (AAGCN)
// theta/phi arrive as one tensor TP (N, 6*Ci, T*V) produced by the 1x1 channel contraction, rows
// ordered [theta_0 | phi_0 | theta_1 | phi_1 | theta_2 | phi_2].
//
// The (c',t) contraction is a 25x25xK product per (sample, subset): K is split over frame tiles
// (one workgroup each, matrix cores), partial VxV tiles go to a slab and the tiny finalize kernel adds
// them in a fixed order, so results are bitwise reproducible.
#include "agcn_common.h"

namespace {

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

constexpr int SC_CK = 16;   // theta/phi channels staged per LDS chunk

__global__ void __launch_bounds__(256)
scores_fwd_kernel(const float* __restrict__ tp, float* __restrict__ spart, int N, int Ci, int T, int V, int tt,
                  int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ttv = tt * V;
  float* Th = smem;                  // [SC_CK*tt][V]
  float* Ph = smem + SC_CK * ttv;    // [SC_CK*tt][V]
  float* red = Ph + SC_CK * ttv;     // [4][V*V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int n = blockIdx.x / ntiles, tile = blockIdx.x - n * ntiles;
  const int i = blockIdx.y;
  const int t0 = tile * tt;
  const int nvalid = min(tt, T - t0) * V;
  const long P = (long)T * V;
  const float* th_base = tp + ((long)n * 6 * Ci + (long)i * 2 * Ci) * P + (long)t0 * V;
  const float* ph_base = th_base + (long)Ci * P;
  f32x16 d;
#pragma unroll
  for (int j = 0; j < 16; ++j) d[j] = 0.f;
  const int lc = min(lr, V - 1);
  for (int c0 = 0; c0 < Ci; c0 += SC_CK) {
    __syncthreads();
    {   // whole chunk fetched before any LDS write: 32 loads in flight per lane (latency-bound kernel)
      float v1[4][SC_CK / 4], v2[4][SC_CK / 4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = lane + 64 * u;
#pragma unroll
        for (int j = 0; j < SC_CK / 4; ++j) {
          const int cl = wave + 4 * j;
          const bool ok = (c0 + cl) < Ci && q < nvalid;
          const long o = (long)(c0 + ((c0 + cl) < Ci ? cl : 0)) * P + (ok ? q : 0);
          v1[u][j] = th_base[o];
          v2[u][j] = ph_base[o];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = lane + 64 * u;
#pragma unroll
        for (int j = 0; j < SC_CK / 4; ++j) {
          const int cl = wave + 4 * j;
          const bool ok = (c0 + cl) < Ci && q < nvalid;
          if (q < ttv) {
            Th[cl * ttv + q] = ok ? v1[u][j] : 0.f;
            Ph[cl * ttv + q] = ok ? v2[u][j] : 0.f;
          }
        }
      }
    }
    __syncthreads();
    const int npairs = (SC_CK * tt) >> 1;   // SC_CK even
    for (int it = wave; it < npairs; it += 4) {
      const int row = 2 * it + h;
      float av = Th[row * V + lc];
      float bv = Ph[row * V + lc];
      av = (lr < V) ? av : 0.f;
      bv = (lr < V) ? bv : 0.f;
      d = mfma32(av, bv, d);
    }
  }
  const int VV = V * V;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int u = mfma_row(j, h);
    if (u < V && lr < V) red[wave * VV + u * V + lr] = d[j];
  }
  __syncthreads();
  float* dst = spart + (((long)n * 3 + i) * ntiles + tile) * VV;
  for (int e = tid; e < VV; e += 256) dst[e] = red[e] + red[VV + e] + red[2 * VV + e] + red[3 * VV + e];
}

// one 64-lane wave per (n, i): sum the partial tiles, scale, column softmax, add the static graph terms
__global__ void __launch_bounds__(256)
adj_finalize_kernel(const float* __restrict__ spart, const float* __restrict__ A, const float* __restrict__ PA,
                    const float* __restrict__ alpha, float* __restrict__ Pout, float* __restrict__ adj, int V,
                    int ntiles, int nused, float inv_k) {
  __shared__ float S[32 * 32];
  const int ni = blockIdx.x, i = ni % 3;
  const int VV = V * V, lane = threadIdx.x;
  const float* src = spart + (long)ni * ntiles * VV;
  for (int e = lane; e < VV; e += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int t = 0; t < nused; ++t) s += src[(long)t * VV + e];      // fixed order; 8 loads in flight
    S[e] = s * inv_k;
  }
  __syncthreads();
  if (lane < V) {
    const int v = lane;
    float mx = -INFINITY;
    for (int u = 0; u < V; ++u) mx = fmaxf(mx, S[u * V + v]);
    float sum = 0.f;
    for (int u = 0; u < V; ++u) {
      const float e = expf(S[u * V + v] - mx);
      S[u * V + v] = e;
      sum += e;
    }
    const float inv = 1.f / sum;
    const float al = alpha ? alpha[0] : 1.f;
    for (int u = 0; u < V; ++u) {
      const float p = S[u * V + v] * inv;
      const int e = u * V + v;
      Pout[(long)ni * VV + e] = p;
      // AGCN: P + A + PA (agcn.py:95,102); AAGCN: PA + alpha*P (aagcn.py:173), A = null
      adj[(long)ni * VV + e] = al * p + (A ? A[i * VV + e] : 0.f) + PA[i * VV + e];
    }
  }
}

// per (n,i): dadj = sum of slot partials; dS = P * (dP - sum_u P dP) / K with dP = alpha*dadj
__global__ void __launch_bounds__(256)
adj_bwd_kernel(const float* __restrict__ dpart, const float* __restrict__ P, const float* __restrict__ alpha,
               float* __restrict__ dadj, float* __restrict__ dS, float* __restrict__ dalpha_part, int V, int nslots,
               float inv_k) {
  __shared__ float D[32 * 32];
  __shared__ float dal[32];
  const int ni = blockIdx.x;
  const int VV = V * V, lane = threadIdx.x;
  const float* src = dpart + (long)ni * nslots * VV;
  for (int e = lane; e < VV; e += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int t = 0; t < nslots; ++t) s += src[(long)t * VV + e];     // fixed order; 8 loads in flight
    D[e] = s;
    dadj[(long)ni * VV + e] = s;
  }
  __syncthreads();
  const float al = alpha ? alpha[0] : 1.f;
  if (lane < V) {
    const int v = lane;
    const float* p = P + (long)ni * VV;
    float dot = 0.f;
    for (int u = 0; u < V; ++u) dot += p[u * V + v] * D[u * V + v];
    for (int u = 0; u < V; ++u) {
      const int e = u * V + v;
      dS[(long)ni * VV + e] = al * p[e] * (D[e] - dot) * inv_k;
    }
    dal[v] = dot;
  }
  if (dalpha_part) {
    __syncthreads();
    if (lane == 0) {
      float s = 0.f;
      for (int v = 0; v < V; ++v) s += dal[v];
      dalpha_part[ni] = s;
    }
  }
}

// dPA[i][e] = sum_n dadj[n][i][e]  (fixed order: four contiguous sample ranges per element, then their sum in range order;
// 64 elements x 4 ranges per block -- the one-thread-per-element loop over all N samples was 31 us of dependent loads)
__global__ void __launch_bounds__(256) dpa_reduce_kernel(const float* __restrict__ dadj, float* __restrict__ dPA, int N,
                                                         int VV3) {
  __shared__ float part[4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int per = (N + 3) >> 2;
  const int n0 = grp * per, n1 = min(N, n0 + per);
  float s = 0.f;
  if (e < VV3) {
#pragma unroll 8
    for (int n = n0; n < n1; ++n) s += dadj[(long)n * VV3 + e];
  }
  part[grp][el] = s;
  __syncthreads();
  if (grp == 0 && e < VV3) dPA[e] = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
}

// dtheta_i[c',t,u] = sum_v dS_i[u,v] phi_i[c',t,v];  dphi_i[c',t,v] = sum_u dS_i[u,v] theta_i[c',t,u]
// (dS already carries the 1/K).  Also per-row sums for the conv_a/conv_b bias gradients.
__global__ void __launch_bounds__(256, 3)
scores_bwd_kernel(const float* __restrict__ tp, const float* __restrict__ dS, float* __restrict__ dtp,
                  float* __restrict__ dbpart, unsigned* __restrict__ amax, int N, int Ci, int T, int V, int tt,
                  int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ unsigned wmax[4];
  unsigned mx = 0;                       // max |dtp| over what this thread stores (bit pattern: unsigned order = float order)
  const int ttv = tt * V;
  const int VS = (V + 1) >> 1, VP = 2 * VS;
  float* Th = smem;                      // [SC_CK*tt][V]
  float* Ph = smem + SC_CK * ttv;
  float* B1 = Ph + SC_CK * ttv;          // [VP][32]: B1[k=v][col=u] = dS[u][v]
  float* B2 = B1 + VP * 32;              // [VP][32]: B2[k=u][col=v] = dS[u][v]
  float* rows = B2 + VP * 32;            // [2][SC_CK*tt] row sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int n = blockIdx.x / ntiles, tile = blockIdx.x - n * ntiles;
  const int i = blockIdx.y;
  const int t0 = tile * tt;
  const int tvalid = min(tt, T - t0);
  const int nvalid = tvalid * V;
  const long P = (long)T * V;
  const long row0 = (long)n * 6 * Ci + (long)i * 2 * Ci;
  const float* dsn = dS + ((long)n * 3 + i) * V * V;
  for (int e = tid; e < VP * 32; e += 256) {
    const int k = e >> 5, col = e & 31;
    const bool ok = k < V && col < V;
    B1[e] = ok ? dsn[col * V + k] : 0.f;
    B2[e] = ok ? dsn[k * V + col] : 0.f;
  }
  const int nrows = SC_CK * tt;
  const int nrt = (nrows + 31) >> 5;
  for (int c0 = 0; c0 < Ci; c0 += SC_CK) {
    __syncthreads();
    // 16-byte loads, lane <-> 4 consecutive positions of a row (tt*V <= 256): 8 vector loads per lane and chunk instead
    // of 32 scalar ones -- the scalar version ran at the pace of the CU's address path, not of HBM.  What a lane reads
    // beyond the tile (next frames of the row) is masked below; only the last rows of the tensor take the clamped path.
    {
      typedef f32x4 f32x4_u __attribute__((aligned(4)));
      const float* tp_end = tp + (long)N * 6 * Ci * P;
      f32x4 v1[SC_CK / 4], v2[SC_CK / 4];
      const int q0 = 4 * lane;
#pragma unroll
      for (int j = 0; j < SC_CK / 4; ++j) {
        const int cl = wave + 4 * j;
        const bool okr = (c0 + cl) < Ci;
        const float* s1 = tp + (row0 + c0 + (okr ? cl : 0)) * P + (long)t0 * V + q0;
        const float* s2 = s1 + (long)Ci * P;
        if (s2 + 4 <= tp_end) {
          v1[j] = *reinterpret_cast<const f32x4_u*>(s1);
          v2[j] = *reinterpret_cast<const f32x4_u*>(s2);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v1[j][e] = (s1 + e < tp_end) ? s1[e] : 0.f;
            v2[j][e] = (s2 + e < tp_end) ? s2[e] : 0.f;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < SC_CK / 4; ++j) {
        const int cl = wave + 4 * j;
        const bool okr = (c0 + cl) < Ci;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = q0 + e;
          if (q < ttv) {
            const bool ok = okr && q < nvalid;
            Th[cl * ttv + q] = ok ? v1[j][e] : 0.f;
            Ph[cl * ttv + q] = ok ? v2[j][e] : 0.f;
          }
        }
      }
    }
    __syncthreads();
    // tiles 0..nrt-1: dtheta from phi rows ; nrt..2nrt-1: dphi from theta rows
    if (2 * nrt <= 12) {
      // Coalesced variant (at most 12 tiles, e.g. V = 25): every wave keeps its (up to 3) result tiles in registers, the results then
      // overwrite the operand tiles in place (tile row r2, column v = LDS element r2*V + v of the [channel][position]
      // image) and leave as whole 1000-byte rows; the bias-gradient row sums are one wave reduction per channel.
      f32x16 dd[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int tl = wave + 4 * k;
#pragma unroll
        for (int j = 0; j < 16; ++j) dd[k][j] = 0.f;
        if (tl < 2 * nrt) {
          const int which = tl / nrt, rt = tl - which * nrt;
          const float* src = which == 0 ? Ph : Th;
          const float* Bf = which == 0 ? B1 : B2;
          const int row = min(rt * 32 + lr, nrows - 1);
          for (int s2 = 0; s2 < VS; ++s2) {
            const int k2 = 2 * s2 + h;
            float av = src[row * V + min(k2, V - 1)];
            av = (k2 < V) ? av : 0.f;
            dd[k] = mfma32(av, Bf[k2 * 32 + lr], dd[k]);
          }
        }
      }
      __syncthreads();                       // every wave is done reading the operand tiles
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int tl = wave + 4 * k;
        if (tl < 2 * nrt) {
          const int which = tl / nrt, rt = tl - which * nrt;
          float* dstb = which == 0 ? Th : Ph;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int r2 = rt * 32 + mfma_row(j, h);
            if (r2 < nrows && lr < V) dstb[r2 * V + lr] = dd[k][j];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float* buf = which == 0 ? Th : Ph;
#pragma unroll
        for (int j = 0; j < SC_CK / 4; ++j) {
          const int cl = wave + 4 * j;
          const bool okr = (c0 + cl) < Ci;
          float* drow = dtp + (row0 + (long)which * Ci + c0 + (okr ? cl : 0)) * P + (long)t0 * V;
          float sum = 0.f;
          {                                    // lane <-> 4 consecutive positions: one 16-byte store when all are valid
            typedef f32x4 f32x4_u __attribute__((aligned(4)));
            const int q = 4 * lane;
            if (okr && q + 3 < nvalid) {
              f32x4 val;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                val[e] = buf[cl * ttv + q + e];
                sum += val[e];
                const float fv = val[e];
                mx = max(mx, __float_as_uint(fv) & 0x7fffffffu);
              }
              *reinterpret_cast<f32x4_u*>(drow + q) = val;
            } else if (okr) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (q + e < nvalid) {
                  const float val = buf[cl * ttv + q + e];
                  drow[q + e] = val;
                  sum += val;
                  mx = max(mx, __float_as_uint(val) & 0x7fffffffu);
                }
            }
          }
          sum = half_sum(sum);
          sum += __shfl_xor(sum, 32);
          if (lane == 0) rows[which * nrows + cl * tt] = sum;       // slot of (channel cl, frame 0); others zeroed below
        }
      }
      // the row-sum table is indexed per (channel, frame): frames 1.. hold nothing in this variant
      for (int e = tid; e < 2 * nrows; e += 256)
        if ((e % nrows) % tt != 0) rows[e] = 0.f;
    } else {
    for (int tl = wave; tl < 2 * nrt; tl += 4) {
        const int which = tl / nrt, rt = tl - which * nrt;
        const float* src = which == 0 ? Ph : Th;
        const float* Bf = which == 0 ? B1 : B2;
        const int row = min(rt * 32 + lr, nrows - 1);
        f32x16 d;
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = 0.f;
        for (int s = 0; s < VS; ++s) {
          const int k = 2 * s + h;
          float av = src[row * V + min(k, V - 1)];
          av = (k < V) ? av : 0.f;
          d = mfma32(av, Bf[k * 32 + lr], d);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int r2 = rt * 32 + mfma_row(j, h);
          const int cl = r2 / tt, t_l = r2 - cl * tt;
          const bool ok = r2 < nrows && (c0 + cl) < Ci && t_l < tvalid;
          const float val = (ok && lr < V) ? d[j] : 0.f;
          mx = max(mx, __float_as_uint(val) & 0x7fffffffu);
          if (ok && lr < V)
            dtp[(row0 + (long)which * Ci + c0 + cl) * P + (long)(t0 + t_l) * V + lr] = val;
          const float rs = half_sum(val);
          if (lr == 0 && r2 < nrows) rows[which * nrows + r2] = rs;
        }
      }
    }
    __syncthreads();
    if (dbpart) {
      for (int e = tid; e < 2 * SC_CK; e += 256) {
        const int which = e / SC_CK, cl = e - which * SC_CK;
        if (c0 + cl < Ci) {
          float s = 0.f;
          for (int t = 0; t < tt; ++t) s += rows[which * nrows + cl * tt + t];
          dbpart[((long)n * ntiles + tile) * 6 * Ci + (long)i * 2 * Ci + (long)which * Ci + c0 + cl] = s;
        }
      }
    }
  }
  if (amax) {
    // one atomic per workgroup at most, none once the running maximum has passed this workgroup's (a stale read only
    // costs a redundant atomic: the value can only grow)
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, k));
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    if (tid == 0) {
      const unsigned m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
      if (m > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, m);
    }
  }
}

// The same gradient, one WAVE per (sample, subset, theta|phi, channel) row block, no barriers.  A row block is T contiguous
// rows of V floats, both in tp and in dtp, and the exact-f32 matrix op wants exactly that shape:
//   dtheta[t, u] = sum_v phi[t, v] dS[u, v]      A = 32 frames x V, B = dS (or its transpose) held in VS registers per lane
//   dphi[t, v]   = sum_u theta[t, u] dS[u, v]    for the wave's life, D = 32 frames x V joints.
// A 32-frame tile is 32 V CONTIGUOUS floats on both sides, so it moves as whole 16-byte pieces (range-checked buffer
// instructions: the ragged last tile needs no masks, what lies past the row block reads as zero and is not written) and
// changes shape in a wave-private piece of LDS: [32][V] rows -> operand A (lane (t, h) reads element 2s + h of its row,
// bank-conflict free for odd V) and accumulators -> [32][V] rows -> 16-byte pieces.  (Round 3's first version read and
// wrote the rows straight from the lanes, 4V-byte pieces of 32 rows per instruction: 3.6 TB/s.)  The next tile's pieces
// are in flight while this tile's VS matrix ops run; the conv_a/conv_b bias gradient of the row block is one wave
// reduction at the end (one slot per sample).
template <int V, int VS>
__global__ void __launch_bounds__(256) scores_bwd_rows_kernel(const float* __restrict__ tp, const float* __restrict__ dS,
                                                              float* __restrict__ dtp, float* __restrict__ dbpart,
                                                              unsigned* __restrict__ amax, int N, int Ci, int T) {
  constexpr int TILE = 32 * V;                         // floats of a tile
  constexpr int NP = (TILE + 255) / 256;               // 16-byte pieces per lane (64 lanes x 4 floats per instruction)
  __shared__ __attribute__((aligned(16))) float lds[4][2][TILE + 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x * 4 + wave;              // ((n * 3 + i) * 2 + which) * Ci + c
  if (unit >= N * 6 * Ci) return;                      // (wave-uniform; no barriers in this kernel)
  const int c = unit % Ci, which = (unit / Ci) & 1, ni = unit / (2 * Ci);
  const int n = ni / 3, i = ni - 3 * n;
  const long P = (long)T * V;
  const long row_th = (long)n * 6 * Ci + (long)i * 2 * Ci + c, row_ph = row_th + Ci;
  const float* src = tp + (which == 0 ? row_ph : row_th) * P;
  float* dst = dtp + (which == 0 ? row_th : row_ph) * P;
  const float* dsn = dS + (long)ni * V * V;
  float* tin = lds[wave][0];
  float* tout = lds[wave][1];
  float bq[VS];
#pragma unroll
  for (int s = 0; s < VS; ++s) {
    const int k = 2 * s + h;
    const bool ok = k < V && lr < V;
    const float v = dsn[ok ? (which == 0 ? lr * V + k : k * V + lr) : 0];
    bq[s] = ok ? v : 0.f;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(P * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(P * 4), 0x00020000);
  // (a piece that straddles the end of the row block goes dword by dword: the range check of a 16-byte access is not
  // relied upon to be per dword)
  auto load_tile = [&](int t0, f32x4 (&pv)[NP]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int f = 4 * (lane + 64 * u);               // float of the tile; pieces past the tile are not needed
      const int o = t0 * V + min(f, TILE - 4);
      const int rem = (int)P - o;
      if (rem >= 4) {
        pv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o * 4, 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pv[u][e] = e < rem ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (o + e) * 4, 0, 0)) : 0.f;
      }
    }
  };
  f32x4 pv[2][NP];
  load_tile(0, pv[0]);
  float bsum = 0.f;
  unsigned mx = 0;
  const int ntile = (T + 31) >> 5;
#pragma unroll 1
  for (int tb = 0; tb < ntile; tb += 2) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t0 = (tb + q) * 32;
      if (t0 < T) {                                    // (wave-uniform)
        if (t0 + 32 < T) load_tile(t0 + 32, pv[q ^ 1]);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
          const int f = 4 * (lane + 64 * u);
          if (f < TILE) *reinterpret_cast<f32x4*>(tin + f) = pv[q][u];
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 d;
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = 0.f;
#pragma unroll
        for (int s = 0; s < VS; ++s) {
          const int k = 2 * s + h;
          float a = tin[lr * V + min(k, V - 1)];
          a = k < V ? a : 0.f;
          d = mfma32(a, bq[s], d);
        }
        if (lr < V) {
#pragma unroll
          for (int j = 0; j < 16; ++j) tout[mfma_row(j, h) * V + lr] = d[j];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NP; ++u) {
          const int f = 4 * (lane + 64 * u);
          if (f < TILE) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(tout + f);
            const int o = t0 * V + f;
            const int rem = (int)P - o;                // floats of this piece inside the row block
            if (rem >= 4) {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, z), rd, o * 4, 0, 0);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float ze = z[e];
                if (e < rem) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ze), rd, (o + e) * 4, 0, 0);
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ze = z[e];
              const float zz = e < rem ? ze : 0.f;       // (rows past the block are not part of the sums)
              bsum += zz;
              mx = max(mx, __float_as_uint(zz) & 0x7fffffffu);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();               // (tout / tin are rewritten by the next tile)
      }
    }
  }
  bsum = half_sum(bsum);
  bsum += __shfl_xor(bsum, 32);
  if (lane == 0) dbpart[which == 0 ? row_th : row_ph] = bsum;       // (N, 6 Ci)
  if (amax) {
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, k));
    if (lane == 0 && mx > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, mx);
  }
}

template <int V, int VS>
int scores_bwd_rows_launch(const float* tp, const float* dS, float* dtp, float* dbpart, unsigned* amax, int N, int Ci, int T,
                           hipStream_t s) {
  const int units = N * 6 * Ci;
  hipLaunchKernelGGL((scores_bwd_rows_kernel<V, VS>), dim3((units + 3) / 4), dim3(256), 0, s, tp, dS, dtp, dbpart, amax, N, Ci,
                     T);
  AGCN_NOTE_KERNEL("scores_bwd_rows_kernel<%d, %d>", V, VS);
  return agcn_check_launch();
}
// AGCN_SCORES_ROWS=0 keeps the LDS-staged kernel (A/B)
static inline bool scores_rows_enabled() {
  static const int on = getenv("AGCN_SCORES_ROWS") ? atoi(getenv("AGCN_SCORES_ROWS")) : 1;
  return on != 0;
}

inline int sc_tile_frames(int V, int T) {
  int tt = 256 / V;
  return tt > T ? T : tt;
}

}  // namespace

// sum the (sample, subset, tile) score slabs, scale by 1/(Ci*T), column softmax, add the static graph terms
// (shared with adj_fused.hip)
// nused: how many of the ntiles slots of each (sample, subset) hold partials (-1: all of them)
int agcn_adj_finalize(const float* spart, const float* A, const float* PA, const float* alpha, float* P, float* adj,
                      int N, int Ci, int T, int V, hipStream_t s, int nused) {
  const int tt = sc_tile_frames(V, T), ntiles = (T + tt - 1) / tt;
  if (nused < 0 || nused > ntiles) nused = ntiles;
  hipLaunchKernelGGL(adj_finalize_kernel, dim3(N * 3), dim3(256), 0, s, spart, A, PA, alpha, P, adj, V, ntiles, nused,
                     1.0f / ((float)Ci * (float)T));
  return agcn_check_launch();
}

extern "C" {

int agcn_scores_num_tiles(int V, int T) {
  int tt = sc_tile_frames(V, T);
  return (T + tt - 1) / tt;
}

// spart: (N,3,ntiles,V,V) scratch ; P, adj: (N,3,V,V).  A may be null (AAGCN), alpha may be null (AGCN).
int agcn_adjacency_fwd(const float* tp, const float* A, const float* PA, const float* alpha, float* spart, float* P,
                       float* adj, int N, int Ci, int T, int V, void* stream) {
  if (!tp || !PA || !spart || !P || !adj || N <= 0 || Ci <= 0 || T <= 0 || V <= 0 || V > 32) return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int tt = sc_tile_frames(V, T), ntiles = (T + tt - 1) / tt;
  const size_t smem = 4 * ((size_t)2 * SC_CK * tt * V + 4 * V * V);
  hipLaunchKernelGGL(scores_fwd_kernel, dim3(N * ntiles, 3), dim3(256), smem, s, tp, spart, N, Ci, T, V, tt, ntiles);
  int rc = agcn_check_launch();
  if (rc) return rc;
  return agcn_adj_finalize(spart, A, PA, alpha, P, adj, N, Ci, T, V, s);
}

// dadj_part: (N,3,nslots,V,V) from agcn_gcn_dadj ; outputs: dadj (N,3,V,V) scratch, dS (N,3,V,V), dPA (3,V,V),
// dalpha_part (N*3) or null
int agcn_adjacency_bwd_softmax(const float* dadj_part, const float* P, const float* alpha, float* dadj, float* dS,
                               float* dPA, float* dalpha_part, int N, int Ci, int T, int V, int nslots,
                               void* stream) {
  if (!dadj_part || !P || !dadj || !dS || !dPA || N <= 0 || V <= 0 || V > 32 || nslots <= 0) return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adj_bwd_kernel, dim3(N * 3), dim3(256), 0, s, dadj_part, P, alpha, dadj, dS, dalpha_part, V,
                     nslots, 1.0f / ((float)Ci * (float)T));
  int rc = agcn_check_launch();
  if (rc) return rc;
  const int VV3 = 3 * V * V;
  hipLaunchKernelGGL(dpa_reduce_kernel, dim3((VV3 + 63) / 64), dim3(256), 0, s, (const float*)dadj, dPA, N, VV3);
  return agcn_check_launch();
}

// dtp: (N,6Ci,T*V) ; dbpart: (N*ntiles, 6Ci) scratch ; scratch: agcn_colsum_scratch_bytes(6Ci) ; db: (6Ci)
int agcn_adjacency_bwd_scores_ex(const float* tp, const float* dS, float* dtp, float* dbpart, void* scratch, float* db,
                                 float* dtp_absmax_out, int N, int Ci, int T, int V, void* stream);
int agcn_adjacency_bwd_scores(const float* tp, const float* dS, float* dtp, float* dbpart, void* scratch, float* db,
                              int N, int Ci, int T, int V, void* stream) {
  return agcn_adjacency_bwd_scores_ex(tp, dS, dtp, dbpart, scratch, db, nullptr, N, Ci, T, V, stream);
}
// dtp_absmax_out (or NULL): device scalar that receives max |dtp| for the f16x3 kernels that read dtp next
int agcn_adjacency_bwd_scores_ex(const float* tp, const float* dS, float* dtp, float* dbpart, void* scratch, float* db,
                                 float* dtp_absmax_out, int N, int Ci, int T, int V, void* stream) {
  if (!tp || !dS || !dtp || !dbpart || !scratch || !db || N <= 0 || Ci <= 0 || T <= 0 || V <= 0 || V > 32)
    return AGCN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int tt = sc_tile_frames(V, T), ntiles = (T + tt - 1) / tt;
  const int VP = 2 * ((V + 1) / 2);
  const size_t smem = 4 * ((size_t)2 * SC_CK * tt * V + 2 * VP * 32 + 2 * SC_CK * tt);
  if (dtp_absmax_out && hipMemsetAsync(dtp_absmax_out, 0, 4, s) != hipSuccess) return AGCN_ERR_ARG;
  if (scores_rows_enabled() && (V == 25 || V == 18) && (long)T * V * 4 < (1L << 31)) {
    unsigned* am = reinterpret_cast<unsigned*>(dtp_absmax_out);
    int rc = V == 25 ? scores_bwd_rows_launch<25, 13>(tp, dS, dtp, dbpart, am, N, Ci, T, s)
                     : scores_bwd_rows_launch<18, 9>(tp, dS, dtp, dbpart, am, N, Ci, T, s);
    if (rc) return rc;
    return agcn_colsum(dbpart, N, 6 * Ci, scratch, db, stream);      // one slot of row sums per sample
  }
  hipLaunchKernelGGL(scores_bwd_kernel, dim3(N * ntiles, 3), dim3(256), smem, s, tp, dS, dtp, dbpart,
                     reinterpret_cast<unsigned*>(dtp_absmax_out), N, Ci, T, V, tt, ntiles);
  int rc = agcn_check_launch();
  if (rc) return rc;
  return agcn_colsum(dbpart, N * ntiles, 6 * Ci, scratch, db, stream);
}

}  // extern "C"
