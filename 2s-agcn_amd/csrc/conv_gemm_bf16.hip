// Split-bf16 ("bf16x6" / "bf16x3") variant of the implicit-GEMM channel contraction of conv_gemm.hip for the plain
// (non-aggregated) temporal convolutions: unit_tcn forward and backward-data (reference agcn.py:40-41,49).
//
// Why: gfx950's f32-input MFMA runs at the vector rate, 1/16 of the bf16 MFMA rate.  Every fp32 operand is split
// exactly into three bf16 pieces  x = hi + mid + lo  (8+8+8 significand bits); the product a*b is then
//   hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi)          [6 bf16 MFMAs, fp32 accumulate]
// which drops only terms below 2^-24 |a*b|: measured error vs fp64 equals that of a plain fp32 GEMM,
// at 6/16 of the fp32 matrix-core time.  NPL=2 keeps (hi,mid) and the 3 leading products ("bf16x3", error ~4e-6).
//
// Layout: weights are pre-split by the pack kernel into per-(row block, K chunk) LDS images
//   A[plane][tap][h][m][8]   (bf16; h = which 8-channel half of the 16-channel K block)
// and the source window is split while it is staged:  B[plane][h][position][8 channels], so that a lane's MFMA
// fragment (8 consecutive k for one row/column) is one conflict-free ds_read_b128, and a temporal tap is a row offset.
#include "agcn_common.h"
#include "epilogue.h"
#include "split_f16.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));     // window rows are only 4-byte aligned

constexpr int CK = 16, NW = 8, NT = 512;

struct BfArgs {
  const float* in;
  const unsigned short* wp;   // packed split weights
  const float* bias;
  float* out;
  float* stats;
  const float* add1;
  const float* mask1;
  const float* add2;
  const float* mask2;
  int N, M, Kinner, in_rows;
  int V, T_src, T_out, tt, ntiles;
  int src_stride, f_off;
  int out_fs, out_fo, T_full;
  int FW, WLR;                 // window frames, window rows (positions) incl. pad
  int accumulate;
  int nchunks, nmb;
  int off_b;                   // byte offset of the B image in LDS
  int off_bias;                // byte offset of the bias row in LDS
  int dbg;                     // profiling switches (AGCN_CB_DBG): 1 = no MFMA loop, 2 = stage chunk 0 only, 4 = no epilogue
  int relu;                    // epilogue: out = max(., 0) (BN-folded inference)
  const float* in_absmax;      // f16x3: max |in| (device scalar written by absmax_kernel): range scaling of the activations
};

struct BfPackArgs {
  const float* w;
  unsigned short* wp;
  int M, Kinner, nchunks;
  long sa_m, sa_c;
  int tap_mul, tap_add, tap_flip_from;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 p = __builtin_convertvector(v, bf16x2);      // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float lo_as_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_as_f32(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// (a, b) -> packed bf16 pairs of the three pieces
__device__ __forceinline__ void split_pair(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
  ph = pack_bf16(a, b);
  const float ra = a - lo_as_f32(ph), rb = b - hi_as_f32(ph);
  pm = pack_bf16(ra, rb);
  pl = pack_bf16(ra - lo_as_f32(pm), rb - hi_as_f32(pm));
}

// ---- "f16x3" (split_f16.h): the temporal convolutions' forward and backward-data run on it, their streamed operand
// range-scaled by the tensor maximum ----
template <bool F16>
__device__ __forceinline__ f32x16 mfma_split(bf16x8 x, bf16x8 y, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
}

// wp[(mb*nchunks + ch)][plane][tap][h][ml][8]
template <int TAPS, int BM, bool F16 = false>
__global__ void __launch_bounds__(256) pack_weights_bf16_kernel(const BfPackArgs p) {
  constexpr int PER_PLANE = TAPS * 2 * BM * 8;
  const int ch = blockIdx.x % p.nchunks, mb = blockIdx.x / p.nchunks;
  unsigned short* dst = p.wp + (long)blockIdx.x * 3 * PER_PLANE;
  for (int e = threadIdx.x; e < PER_PLANE / 2; e += 256) {      // one pair (j, j+1) per thread-iteration
    const int j = (e & 3) * 2;
    const int ml = (e >> 2) % BM;
    const int r = (e >> 2) / BM;
    const int h = r & 1, tap = r >> 1;
    const int m = mb * BM + ml;
    const int gt = p.tap_flip_from >= 0 ? (p.tap_flip_from - (tap * p.tap_mul + p.tap_add)) : tap;
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kc = ch * CK + h * 8 + j + q;
      v[q] = (m < p.M && kc < p.Kinner) ? p.w[(long)m * p.sa_m + (long)kc * p.sa_c + gt] : 0.f;
    }
    unsigned ph, pm, pl = 0;
    if constexpr (F16) split_pair_f16(v[0] * F16_W_SCALE, v[1] * F16_W_SCALE, ph, pm);
    else split_pair(v[0], v[1], ph, pm, pl);
    const int o = ((tap * 2 + h) * BM + ml) * 8 + j;
    *reinterpret_cast<unsigned*>(dst + 0 * PER_PLANE + o) = ph;
    *reinterpret_cast<unsigned*>(dst + 1 * PER_PLANE + o) = pm;
    *reinterpret_cast<unsigned*>(dst + 2 * PER_PLANE + o) = pl;
  }
}

// NPL = 3: bf16x6 (fp32-equivalent) ; NPL = 2: bf16x3.  WQ = 64-row blocks per staging quarter (2: windows <= 512 rows)
// TN = 32-position tiles per wave (2: 64-row blocks cover 512 positions, so every weight fragment feeds two MFMAs and
// the 8-frame halo of the window is amortised over 20 frames instead of 10)
template <int TAPS, int NPL, int WQ, int TM, int TN = 1, bool F16 = false>
__global__ void __launch_bounds__(NT, (TAPS == 1 && TM == 2 && TN == 1) ? 4 : 2) conv_gemm_bf16_kernel(const BfArgs a) {
  constexpr int BM = TM * 32;
  constexpr int A_PLANE = TAPS * 2 * BM * 16;          // bytes per plane of the A image
  constexpr int A4 = NPL * A_PLANE / 16;               // 16-byte units of the A image that are used
  constexpr int EA = (A4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ab = smem;
  unsigned char* Bb = smem + a.off_b;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mbk = bid % a.nmb;
  const int nt_id = bid / a.nmb;
  const int n = nt_id / a.ntiles, tile_id = nt_id - n * a.ntiles;
  const int m0 = mbk * BM;
  const int V = a.V, tt = a.tt, t0 = tile_id * tt;
  const int ttv = tt * V;
  const int nvalid = min(tt, a.T_out - t0) * V;
  const int Psrc = a.T_src * V;
  const int f0 = t0 * a.src_stride + a.f_off;
  const int WL = a.FW * V, WLR = a.WLR;
  const int g0 = f0 * V;
  float rs_s = 1.f, rs_inv = 1.f;                      // f16x3 range scale of the activations
  if constexpr (F16) { f16_range_scale(a.in_absmax, rs_s, rs_inv); rs_inv *= F16_W_INV; }

  int boff[TN];                           // this wave's TN x 32 positions
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int q = (wave * TN + tn) * 32 + lr;
    if (q >= ttv) q = 0;
    const int tl = q / V;
    boff[tn] = tl * a.src_stride * V + (q - tl * V);
  }

  float* bias_s = reinterpret_cast<float*>(smem + a.off_bias);   // beyond everything the epilogue tile overwrites
  for (int e = tid; e < BM; e += NT) bias_s[e] = (a.bias && m0 + e < a.M) ? a.bias[m0 + e] : 0.f;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[tm][tn][j] = 0.f;

  // staging roles: wave -> (channel half hb, window quarter wq)
  const int hb = wave & 1, wq = wave >> 1;
  const int qlen = (WL + 3) >> 2;          // rows per quarter
  u32x4 ra[EA];
  float rb[WQ][8];
  const u32x4* wp4 = reinterpret_cast<const u32x4*>(a.wp) + (long)mbk * a.nchunks * (3 * A_PLANE / 16);

  auto issue_loads = [&](int ch) __attribute__((always_inline)) {
    const u32x4* src = wp4 + (long)ch * (3 * A_PLANE / 16);
#pragma unroll
    for (int u = 0; u < EA; ++u) ra[u] = src[min(tid + u * NT, A4 - 1)];
    const int kc0 = ch * CK + hb * 8;
#pragma unroll
    for (int u = 0; u < WQ; ++u) {
      const int rr = lane + 64 * u;
      const int r = wq * qlen + rr;
      const int gp = g0 + r;
      const bool okp = rr < qlen && r < WL && gp >= 0 && gp < Psrc;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bool ok = okp && (kc0 + c) < a.Kinner;
        rb[u][c] = a.in[((long)n * a.in_rows + (ok ? (kc0 + c) : 0)) * Psrc + (ok ? gp : 0)];
      }
    }
  };
  auto commit_lds = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < EA; ++u)
      if (tid + u * NT < A4) reinterpret_cast<u32x4*>(Ab)[tid + u * NT] = ra[u];
    const int kc0 = ch * CK + hb * 8;
#pragma unroll
    for (int u = 0; u < WQ; ++u) {
      const int rr = lane + 64 * u;
      const int r = wq * qlen + rr;
      const int gp = g0 + r;
      const bool okp = rr < qlen && r < WL && gp >= 0 && gp < Psrc;
      u32x4 ph, pm, pl;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float x0 = (okp && (kc0 + 2 * c) < a.Kinner) ? rb[u][2 * c] : 0.f;
        const float x1 = (okp && (kc0 + 2 * c + 1) < a.Kinner) ? rb[u][2 * c + 1] : 0.f;
        unsigned a0, a1, a2 = 0;
        if constexpr (F16) split_pair_f16(x0 * rs_s, x1 * rs_s, a0, a1);
        else split_pair(x0, x1, a0, a1, a2);
        ph[c] = a0; pm[c] = a1; pl[c] = a2;
      }
      if (rr < qlen && r < WL) {
        *reinterpret_cast<u32x4*>(Bb + ((0 * 2 + hb) * WLR + r) * 16) = ph;
        if (NPL >= 2) *reinterpret_cast<u32x4*>(Bb + ((1 * 2 + hb) * WLR + r) * 16) = pm;
        if (NPL == 3) *reinterpret_cast<u32x4*>(Bb + ((2 * 2 + hb) * WLR + r) * 16) = pl;
      }
    }
  };

  const int nchunks = a.nchunks;
  if (nchunks > 0) issue_loads(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
    if (!(a.dbg & 2) || ch == 0) {
      commit_lds(ch);
      if (ch + 1 < nchunks) issue_loads(ch + 1);
    }
    __syncthreads();
    if (a.dbg & 1) continue;
    // the fragments of the next (tap, row tile) step are read from LDS while this step's MFMAs run
    bf16x8 afc[NPL], bfc[NPL][TN];
    auto load_a = [&](bf16x8 (&af)[NPL], int tap, int tm) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        af[pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_PLANE + (((tap * 2 + h) * BM) + tm * 32 + lr) * 16);
    };
    auto load_b = [&](bf16x8 (&bf)[NPL][TN], int tap) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          bf[pl][tn] = *reinterpret_cast<const bf16x8*>(Bb + ((pl * 2 + h) * WLR + boff[tn] + tap * V) * 16);
    };
    load_b(bfc, 0);
    load_a(afc, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NPL * (TN + 1), 0);
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      bf16x8 bfn[NPL][TN];
      const int tapn = min(tap + 1, TAPS - 1);       // the last tap re-reads its own fragments (unused)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        bf16x8 afn[NPL];
        load_a(afn, tm + 1 == TM ? tapn : tap, tm + 1 == TM ? 0 : tm + 1);
        if (tm + 1 == TM) load_b(bfn, tapn);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          // smallest products first
          if (NPL == 3) {
            acc[tm][tn] = mfma_split<F16>(afc[NPL - 1], bfc[0][tn], acc[tm][tn]);
            acc[tm][tn] = mfma_split<F16>(afc[0], bfc[NPL - 1][tn], acc[tm][tn]);
            acc[tm][tn] = mfma_split<F16>(afc[NPL >= 2 ? 1 : 0], bfc[NPL >= 2 ? 1 : 0][tn], acc[tm][tn]);
          }
          if (NPL >= 2) {
            acc[tm][tn] = mfma_split<F16>(afc[NPL >= 2 ? 1 : 0], bfc[0][tn], acc[tm][tn]);
            acc[tm][tn] = mfma_split<F16>(afc[0], bfc[NPL >= 2 ? 1 : 0][tn], acc[tm][tn]);
          }
          acc[tm][tn] = mfma_split<F16>(afc[0], bfc[0][tn], acc[tm][tn]);
        }
        // pin the order "reads of the next step, then this step's MFMAs" (the scheduler otherwise sinks the reads to
        // their uses and waits for each)
        __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
        if (tm + 1 == TM) __builtin_amdgcn_sched_group_barrier(0x100, NPL * TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TN * (NPL == 3 ? 6 : NPL == 2 ? 3 : 1), 0);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) afc[pl] = afn[pl];
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bfc[pl][tn] = bfn[pl][tn];
    }
  }

  // ---- epilogue (epilogue.h): tile -> LDS -> coalesced row stores, residual operands, (sum, sumsq) partials ----
  if ((a.dbg & 4) && acc[0][0][0] != 12345.f) return;
  __syncthreads();                                     // every wave is done with the A/B images
  float* tile = reinterpret_cast<float*>(smem);        // [BM][TP]
  const int TP = NW * TN * 32 + 1;
  float* red = tile + BM * TP;                         // [NT * 2]
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int j = 0; j < 16; ++j)
        tile[(tm * 32 + mfma_row(j, h)) * TP + (wave * TN + tn) * 32 + lr] = F16 ? acc[tm][tn][j] * rs_inv : acc[tm][tn][j];
  __syncthreads();
  int poff[4 * TN];
#pragma unroll
  for (int u = 0; u < 4 * TN; ++u) {
    const int q2 = lane + 64 * u;
    const int tl2 = q2 / V;
    poff[u] = tl2 * a.out_fs * V + (q2 - tl2 * V);
  }
  EpiPtrs ep;
  ep.out = a.out; ep.add1 = a.add1; ep.mask1 = a.mask1; ep.add2 = a.add2; ep.mask2 = a.mask2; ep.stats = a.stats;
  ep.accumulate = a.accumulate; ep.relu = a.relu;
  const long Pfull = (long)a.T_full * V;
  const long rows0 = (long)n * a.M * Pfull + ((long)t0 * a.out_fs + a.out_fo) * V;
  epilogue_rows<BM, NW, 4 * TN>(ep, tile, TP, bias_s, red, a.M, m0, rows0, Pfull, nvalid, poff, (long)n * a.ntiles + tile_id);
}


// ---------------------------------------------------------------------------------------------------------------------
// Producer / consumer variant with a weight ring (128-row blocks, stride-1 windows, K a multiple of 16).
// Measured on the kernel above (l9 shape): 0.20 of 1.44 ms is staging that no MFMA overlaps (all eight waves write the
// 110 KB weight image + the split window between two barriers), another 0.18 the matrix loop's own stalls.  Here
//   * 8 consumer waves only read fragments and issue MFMAs: wave w owns 32 positions x 128 rows, one (chunk, tap)
//     step = 24 MFMAs, the fragments of step g+1 are read while step g runs;
//   * 4 producer waves feed them: the packed weight image of one (chunk, tap) step is ONE contiguous 12 KB slot that
//     goes global -> LDS by LDS-DMA (no registers, no VALU, no ds_write) into a ring of R slots, R-2 steps ahead of its
//     use; the source window of the next 16-channel chunk is loaded 16 bytes per lane, split into bf16 planes and
//     written to the other of two window buffers.
// Only the producers have vector-memory operations in flight, so the consumers' LDS reads never wait on a vmcnt.
// One barrier per step.  The ring slot of step g is (g mod R).
// F16: the planes are fp16 pieces (NPL = 2, three products): forward convolution only
template <int TAPS, int NPL, int R, int NWP = 4, bool F16 = false>
__global__ void __launch_bounds__((8 + NWP) * 64, 3) conv_pc_kernel(const BfArgs a) {
  constexpr int TM = 4, BM = 128, NWC = 8, NTALL = (NWC + NWP) * 64, NTP = NWP * 64;
  constexpr int SLOT = NPL * 2 * BM * 16;               // bytes of one step's weight image [plane][h][m][8]
  constexpr int PIECES = SLOT / 1024;                   // 1 KB LDS-DMA pieces per slot
  constexpr int PPW = (PIECES + NWP - 1) / NWP;         // pieces per producer wave
  constexpr int NMF = NPL == 3 ? 6 : NPL == 2 ? 3 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  const int WLR = a.WLR, V = a.V;
  const int B_BYTES = NPL * 2 * WLR * 16;
  unsigned char* Bbase = smem + a.off_b;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mbk = bid % a.nmb;
  const int nt_id = bid / a.nmb;
  const int n = nt_id / a.ntiles, tile_id = nt_id - n * a.ntiles;
  const int m0 = mbk * BM;
  const int tt = a.tt, t0 = tile_id * tt;
  const int ttv = tt * V;
  const int nvalid = min(tt, a.T_out - t0) * V;
  const int Psrc = a.T_src * V;
  const int g0 = (t0 + a.f_off) * V;                    // first window position (may be negative: zero padding)
  const int WL = a.FW * V;
  const int nchunks = a.nchunks;
  const int G = nchunks * TAPS;
  float rs_s = 1.f, rs_inv = 1.f;                      // f16x3 range scale of the activations
  if constexpr (F16) { f16_range_scale(a.in_absmax, rs_s, rs_inv); rs_inv *= F16_W_INV; }

  float* bias_s = reinterpret_cast<float*>(smem + a.off_bias);
  for (int e = threadIdx.x; e < BM; e += NTALL) bias_s[e] = (a.bias && m0 + e < a.M) ? a.bias[m0 + e] : 0.f;

  auto step_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  if (wave >= NWC) {
    // =============================================== producers ===============================================
    const int pw = wave - NWC;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wp) + (long)mbk * G * SLOT + lane * 16;
    auto dma_slot = [&](int g) __attribute__((always_inline)) {   // step g -> ring slot g % R (all lanes, uniform g)
      const unsigned char* src = wsrc + (long)g * SLOT;
      unsigned char* dst = ring + (g % R) * SLOT;
#pragma unroll
      for (int u = 0; u < PPW; ++u) {
        const int piece = pw + u * NWP;
        if (piece < PIECES)
          __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 1024), (lptr_t)(dst + piece * 1024), 16, 0, 0);
      }
    };
    // window staging: lane task = (channel half hb, 4 consecutive window rows): 2 * ceil(WL / 4) tasks over 256 threads
    const int ptid = threadIdx.x - NWC * 64;
    const int npiece = (WL + 3) >> 2;
    const int ntask = 2 * npiece;
    constexpr int TK = 256 / NTP;                        // tasks per thread (2 * 128 pieces: windows up to 512 rows)
    f32x4 rb[TK][8];
    auto issue_B = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < TK; ++k) {
        if (k * NTP >= ntask) continue;                  // uniform
        int task = ptid + k * NTP;
        if (task >= ntask) task = ntask - 1;             // duplicates rewrite identical data
        const int hb = task & 1, r0 = (task >> 1) * 4;
        const int gp = g0 + r0;
        const float* src = a.in + ((long)n * a.in_rows + ch * CK + hb * 8) * Psrc + gp;
        if (gp >= 0 && gp + 3 < Psrc) {
#pragma unroll
          for (int c = 0; c < 8; ++c) rb[k][c] = *reinterpret_cast<const f32x4_u*>(src + (long)c * Psrc);
        } else {                                          // sample edge: the temporal zero padding
#pragma unroll
          for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool ok = gp + e >= 0 && gp + e < Psrc;
              const float t = src[(long)c * Psrc + (ok ? e : -gp)];
              rb[k][c][e] = ok ? t : 0.f;
            }
        }
      }
    };
    auto commit_B = [&](unsigned char* Bb) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < TK; ++k) {
        if (k * NTP >= ntask) continue;
        int task = ptid + k * NTP;
        if (task >= ntask) task = ntask - 1;
        const int hb = task & 1, r0 = (task >> 1) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          u32x4 ph, pm, pl;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            unsigned q0, q1, q2 = 0;
            if constexpr (F16) split_pair_f16(rb[k][2 * c][e] * rs_s, rb[k][2 * c + 1][e] * rs_s, q0, q1);
            else split_pair(rb[k][2 * c][e], rb[k][2 * c + 1][e], q0, q1, q2);
            ph[c] = q0; pm[c] = q1; pl[c] = q2;
          }
          const int r = r0 + e;
          if (r < WLR) {
            *reinterpret_cast<u32x4*>(Bb + ((0 * 2 + hb) * WLR + r) * 16) = ph;
            if (NPL >= 2) *reinterpret_cast<u32x4*>(Bb + ((1 * 2 + hb) * WLR + r) * 16) = pm;
            if (NPL == 3) *reinterpret_cast<u32x4*>(Bb + ((2 * 2 + hb) * WLR + r) * 16) = pl;
          }
        }
      }
    };
    // prologue: ring slots of steps 0 .. R-2, window of chunk 0
#pragma unroll
    for (int g = 0; g < R - 1; ++g)
      if (g < G) dma_slot(g);
    issue_B(0);
    commit_B(Bbase);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    step_barrier();
    int g = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap, ++g) {
        if (a.dbg & 2) { step_barrier(); continue; }
        if (g + R - 1 < G && !(a.dbg & 8)) dma_slot(g + R - 1);   // into the slot of step g-1, free since the last barrier
        if (TAPS >= 3) {
          if (tap == 0 && ch + 1 < nchunks && !(a.dbg & 4)) issue_B(ch + 1);
          if (tap == (TAPS >= 6 ? TAPS - 3 : TAPS - 2) && ch + 1 < nchunks && !(a.dbg & 4))
            commit_B(Bbase + ((ch + 1) & 1) * B_BYTES);  // loads had TAPS-3 steps to land; readers start at tap TAPS-1
        } else {
          if (tap == 0 && ch + 1 < nchunks) { issue_B(ch + 1); commit_B(Bbase + ((ch + 1) & 1) * B_BYTES); }
        }
        // the slot of step g+2 is read from the next barrier on: at most the R-3 NEWER slots may still be in flight.
        // (vector-memory operations complete in order, so "at most PPW*(R-3) outstanding" implies slot g+2 has landed
        // only while newer slots keep being issued; in the last R-1 steps nothing newer exists: wait for everything)
        if (g + R - 1 < G) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW * (R - 3)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step_barrier();
      }
    }
    return;
  }

  // ================================================= consumers =================================================
  const int lr = lane & 31, h = lane >> 5;
  int q = wave * 32 + lr;
  if (q >= ttv) q = 0;
  const int boff = q;                                    // stride-1 window: position q of the tile = window row q (+tap*V)
  f32x16 acc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;

  const unsigned char* a_lane = ring + (h * BM + lr) * 16;
  auto load_a = [&](bf16x8 (&af)[NPL], int slot, int tm) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      af[pl] = *reinterpret_cast<const bf16x8*>(a_lane + slot * SLOT + (pl * 2 * BM + tm * 32) * 16);
  };
  auto load_b = [&](bf16x8 (&bf)[NPL], const unsigned char* Bb, int tap) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      bf[pl] = *reinterpret_cast<const bf16x8*>(Bb + ((pl * 2 + h) * WLR + boff + tap * V) * 16);
  };
  step_barrier();                                        // ring slots 0 .. R-2 and window 0 are in place
  bf16x8 afc[NPL], bfc[NPL];
  load_b(bfc, Bbase, 0);
  load_a(afc, 0, 0);
  int slot = 0;                                          // g % R
  for (int ch = 0; ch < nchunks; ++ch) {
    const unsigned char* Bb = Bbase + (ch & 1) * B_BYTES;
    const unsigned char* Bn = Bbase + ((ch + 1) & 1) * B_BYTES;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int slot_n = (slot + 1 == R) ? 0 : slot + 1;
      if (a.dbg & 1) { step_barrier(); continue; }
      bf16x8 bfn[NPL];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        bf16x8 afn[NPL];
        if (tm + 1 < TM) {
          load_a(afn, slot, tm + 1);
        } else {
          // first fragments of the next step (the last step re-reads a valid slot; unused)
          load_a(afn, slot_n, 0);
          if (tap + 1 < TAPS) load_b(bfn, Bb, tap + 1);
          else load_b(bfn, Bn, 0);
        }
        if (NPL == 3) {
          acc[tm] = mfma_split<F16>(afc[NPL - 1], bfc[0], acc[tm]);
          acc[tm] = mfma_split<F16>(afc[0], bfc[NPL - 1], acc[tm]);
          acc[tm] = mfma_split<F16>(afc[NPL >= 2 ? 1 : 0], bfc[NPL >= 2 ? 1 : 0], acc[tm]);
        }
        if (NPL >= 2) {
          acc[tm] = mfma_split<F16>(afc[NPL >= 2 ? 1 : 0], bfc[0], acc[tm]);
          acc[tm] = mfma_split<F16>(afc[0], bfc[NPL >= 2 ? 1 : 0], acc[tm]);
        }
        acc[tm] = mfma_split<F16>(afc[0], bfc[0], acc[tm]);
        __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
        if (tm + 1 == TM) __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) afc[pl] = afn[pl];
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) bfc[pl] = bfn[pl];
      slot = slot_n;
      step_barrier();
    }
  }

  // ---- epilogue (epilogue.h), consumers only: the producers have left, the barriers count the live waves ----
  float* tile = reinterpret_cast<float*>(smem);        // [BM][TP]
  const int TP = NWC * 32 + 1;
  float* red = tile + BM * TP;                         // [512 * 2]
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[(tm * 32 + mfma_row(j, h)) * TP + wave * 32 + lr] = F16 ? acc[tm][j] * rs_inv : acc[tm][j];
  __syncthreads();
  int poff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int q2 = lane + 64 * u;
    const int tl2 = q2 / V;
    poff[u] = tl2 * a.out_fs * V + (q2 - tl2 * V);
  }
  EpiPtrs ep;
  ep.out = a.out; ep.add1 = a.add1; ep.mask1 = a.mask1; ep.add2 = a.add2; ep.mask2 = a.mask2; ep.stats = a.stats;
  ep.accumulate = a.accumulate; ep.relu = a.relu;
  const long Pfull = (long)a.T_full * V;
  const long rows0 = (long)n * a.M * Pfull + ((long)t0 * a.out_fs + a.out_fo) * V;
  epilogue_rows<BM, NWC, 4>(ep, tile, TP, bias_s, red, a.M, m0, rows0, Pfull, nvalid, poff, (long)n * a.ntiles + tile_id);
}

// slot-major weight images for conv_pc_kernel: wp[(mb*nchunks + ch)*TAPS + tap][plane][h][ml][8]
template <int TAPS, int BM, int NPL, bool F16 = false>
__global__ void __launch_bounds__(256) pack_weights_slots_kernel(const BfPackArgs p) {
  constexpr int PLANE = 2 * BM * 8;                    // bf16 elements of one plane of a slot
  const int tap = blockIdx.x % TAPS;
  const int ch = (blockIdx.x / TAPS) % p.nchunks, mb = blockIdx.x / (TAPS * p.nchunks);
  unsigned short* dst = p.wp + (long)blockIdx.x * NPL * PLANE;
  const int gt = p.tap_flip_from >= 0 ? (p.tap_flip_from - (tap * p.tap_mul + p.tap_add)) : tap;
  for (int e = threadIdx.x; e < PLANE / 2; e += 256) {
    const int j = (e & 3) * 2;
    const int ml = (e >> 2) % BM;
    const int h = (e >> 2) / BM;
    const int m = mb * BM + ml;
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kc = ch * CK + h * 8 + j + q;
      v[q] = (m < p.M && kc < p.Kinner) ? p.w[(long)m * p.sa_m + (long)kc * p.sa_c + gt] : 0.f;
    }
    unsigned ph, pm, pl = 0;
    if constexpr (F16) split_pair_f16(v[0] * F16_W_SCALE, v[1] * F16_W_SCALE, ph, pm);
    else split_pair(v[0], v[1], ph, pm, pl);
    const int o = (h * BM + ml) * 8 + j;
    *reinterpret_cast<unsigned*>(dst + 0 * PLANE + o) = ph;
    if (NPL >= 2) *reinterpret_cast<unsigned*>(dst + 1 * PLANE + o) = pm;
    if (NPL == 3) *reinterpret_cast<unsigned*>(dst + 2 * PLANE + o) = pl;
  }
}

struct BfGeom {
  int tt, ntiles, FW, WLR, nchunks, nmb, off_b, off_bias;
  size_t smem_bytes, pack_bytes;
};

template <int TAPS, int BM, int TN = 1>
BfGeom bf_geometry(int V, int T_out, int src_stride, int M, int Kinner) {
  BfGeom g;
  g.tt = (256 * TN) / V;
  if (g.tt > T_out) g.tt = T_out;
  if (g.tt < 1) g.tt = 1;
  g.ntiles = (T_out + g.tt - 1) / g.tt;
  g.FW = (g.tt - 1) * src_stride + TAPS;
  g.WLR = g.FW * V + 8;
  g.nchunks = (Kinner + CK - 1) / CK;
  g.nmb = (M + BM - 1) / BM;
  const size_t a_bytes = (size_t)3 * TAPS * 2 * BM * 16;
  g.off_b = (int)a_bytes;
  const size_t b_bytes = (size_t)3 * 2 * g.WLR * 16;
  size_t main_b = a_bytes + b_bytes;
  size_t epi_b = (size_t)BM * (NW * TN * 32 + 1) * 4 + (size_t)NT * 2 * 4;
  g.smem_bytes = ((main_b > epi_b ? main_b : epi_b) + 15) & ~(size_t)15;
  g.off_bias = (int)g.smem_bytes;
  g.smem_bytes += (size_t)BM * 4;
  g.pack_bytes = (size_t)g.nmb * g.nchunks * a_bytes;
  return g;
}

struct BfProblem {
  BfArgs a;
  int fwd_f16;                 // use the f16x3 arithmetic (a.in_absmax holds the streamed operand's max)
  const float* w;
  long sa_m, sa_c;
  int tap_mul, tap_add, tap_flip_from;
  void* ws;
  size_t ws_bytes;
};

template <int TAPS, int NPL, int WQ, int TM, int TN = 1, bool F16 = false>
int launch_bf(BfProblem& p, hipStream_t stream) {
  constexpr int BM = TM * 32;
  BfArgs a = p.a;
  const BfGeom g = bf_geometry<TAPS, BM, TN>(a.V, a.T_out, a.src_stride, a.M, a.Kinner);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if ((g.FW * a.V + 3) / 4 > WQ * 64) return AGCN_ERR_UNSUPPORTED;
  if (g.pack_bytes > p.ws_bytes) return AGCN_ERR_WORKSPACE;
  a.tt = g.tt; a.ntiles = g.ntiles; a.FW = g.FW; a.WLR = g.WLR; a.nchunks = g.nchunks; a.nmb = g.nmb;
  a.off_b = g.off_b; a.off_bias = g.off_bias;
  a.wp = (const unsigned short*)p.ws;
  static const int dbg = getenv("AGCN_CB_DBG") ? atoi(getenv("AGCN_CB_DBG")) : 0;
  a.dbg = dbg;
  if (g.nchunks > 0) {
    BfPackArgs pk;
    pk.w = p.w; pk.wp = (unsigned short*)p.ws; pk.M = a.M; pk.Kinner = a.Kinner; pk.nchunks = g.nchunks;
    pk.sa_m = p.sa_m; pk.sa_c = p.sa_c;
    pk.tap_mul = p.tap_mul; pk.tap_add = p.tap_add; pk.tap_flip_from = p.tap_flip_from;
    hipLaunchKernelGGL((pack_weights_bf16_kernel<TAPS, BM, F16>), dim3(g.nmb * g.nchunks), dim3(256), 0, stream, pk);
    int rc = agcn_check_launch();
    if (rc) return rc;
  }
  auto kern = conv_gemm_bf16_kernel<TAPS, NPL, WQ, TM, TN, F16>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};   // per (kernel instantiation, device): the attribute is per device
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  AGCN_NOTE_KERNEL("conv_gemm_bf16_kernel<%d, %d, %d, %d, %d, %s>", TAPS, NPL, WQ, TM, TN, F16 ? "true" : "false");
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * g.ntiles * g.nmb)), dim3(NT), g.smem_bytes, stream, a);
  return agcn_check_launch();
}


// ---- producer / consumer ring variant: geometry + launch ----
template <int TAPS, int NPL, int R>
BfGeom bf_geometry_pc(int V, int T_out, int M, int Kinner) {
  constexpr int BM = 128, SLOT = NPL * 2 * BM * 16;
  BfGeom g;
  g.tt = 256 / V;
  if (g.tt > T_out) g.tt = T_out;
  if (g.tt < 1) g.tt = 1;
  g.ntiles = (T_out + g.tt - 1) / g.tt;
  g.FW = (g.tt - 1) + TAPS;
  g.WLR = g.FW * V + 8;
  g.nchunks = (Kinner + CK - 1) / CK;
  g.nmb = (M + BM - 1) / BM;
  g.off_b = R * SLOT;
  const size_t main_b = (size_t)g.off_b + 2 * (size_t)NPL * 2 * g.WLR * 16;
  const size_t epi_b = (size_t)BM * (8 * 32 + 1) * 4 + (size_t)512 * 2 * 4;
  g.smem_bytes = ((main_b > epi_b ? main_b : epi_b) + 15) & ~(size_t)15;
  g.off_bias = (int)g.smem_bytes;
  g.smem_bytes += (size_t)BM * 4;
  g.pack_bytes = (size_t)g.nmb * g.nchunks * TAPS * SLOT;
  return g;
}

// AGCN_CONV_PC=0 selects the single-role kernel (A/B measurement)
static inline bool conv_pc_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AGCN_CONV_PC");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v == 1;
}

template <int TAPS, int NPL, int R = 5, int NWP = 4, bool F16 = false>
int launch_pc(BfProblem& p, hipStream_t stream) {
  constexpr int BM = 128;
  BfArgs a = p.a;
  const BfGeom g = bf_geometry_pc<TAPS, NPL, R>(a.V, a.T_out, a.M, a.Kinner);
  if (g.smem_bytes > 160 * 1024) return AGCN_ERR_UNSUPPORTED;
  if (g.pack_bytes > p.ws_bytes) return AGCN_ERR_WORKSPACE;
  a.tt = g.tt; a.ntiles = g.ntiles; a.FW = g.FW; a.WLR = g.WLR; a.nchunks = g.nchunks; a.nmb = g.nmb;
  a.off_b = g.off_b; a.off_bias = g.off_bias;
  a.wp = (const unsigned short*)p.ws;
  BfPackArgs pk;
  pk.w = p.w; pk.wp = (unsigned short*)p.ws; pk.M = a.M; pk.Kinner = a.Kinner; pk.nchunks = g.nchunks;
  pk.sa_m = p.sa_m; pk.sa_c = p.sa_c;
  pk.tap_mul = p.tap_mul; pk.tap_add = p.tap_add; pk.tap_flip_from = p.tap_flip_from;
  hipLaunchKernelGGL((pack_weights_slots_kernel<TAPS, BM, NPL, F16>), dim3(g.nmb * g.nchunks * TAPS), dim3(256), 0, stream, pk);
  int rc = agcn_check_launch();
  if (rc) return rc;
  auto kern = conv_pc_kernel<TAPS, NPL, R, NWP, F16>;
  static unsigned char lds_ok[AGCN_MAX_DEVICES] = {};
  if (int e = agcn_allow_big_lds_rt(reinterpret_cast<const void*>(kern), lds_ok)) return e;
  static const int dbg = getenv("AGCN_CB_DBG") ? atoi(getenv("AGCN_CB_DBG")) : 0;
  a.dbg = dbg;
  AGCN_NOTE_KERNEL("conv_pc_kernel<%d, %d, %d, %d, %s>", TAPS, NPL, R, NWP, F16 ? "true" : "false");   // (as rocprofv3 prints it)
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * g.ntiles * g.nmb)), dim3((8 + NWP) * 64), g.smem_bytes, stream, a);
  return agcn_check_launch();
}

template <int TAPS>
bool pc_applies(const BfProblem& p, int npl) {
  if (!conv_pc_enabled() || TAPS < 3 || npl != 3) return false;
  if (p.a.M % 128 != 0 || p.a.Kinner % CK != 0 || p.a.src_stride != 1 || p.a.V > 32) return false;
  const BfGeom g = bf_geometry_pc<TAPS, 3, 5>(p.a.V, p.a.T_out, p.a.M, p.a.Kinner);
  return g.smem_bytes <= 160 * 1024 && g.FW * p.a.V <= 512;
}

// 128-row blocks (4 m-tiles per wave) when M allows it: the staged+split window is reused by twice the MFMAs
template <int TAPS, int WQ>
int launch_npl(BfProblem& p, int npl, hipStream_t s) {
  const bool fits128 = bf_geometry<TAPS, 128>(p.a.V, p.a.T_out, p.a.src_stride, p.a.M, p.a.Kinner).smem_bytes <=
                       160 * 1024;
  if constexpr (TAPS >= 3) {
    if (pc_applies<TAPS>(p, npl)) {
      static const int nwp2 = getenv("AGCN_CONV_NWP2") ? atoi(getenv("AGCN_CONV_NWP2")) : 0;
      if (nwp2) return launch_pc<TAPS, 3, 5, 2>(p, s);
      if (p.fwd_f16) return launch_pc<TAPS, 2, 5, 4, true>(p, s);
      return launch_pc<TAPS, 3>(p, s);
    }
  }
  if constexpr (TAPS >= 3) {
    if (p.fwd_f16) {      // f16x3 (three fp16 products), same tiles
      if (p.a.M % 128 == 0 && fits128) return launch_bf<TAPS, 2, WQ, 4, 1, true>(p, s);
      if (agcn_bf16_conv_wide(TAPS, p.a.M)) {
        const int rc = launch_bf<TAPS, 2, 3, 2, 2, true>(p, s);
        if (rc != AGCN_ERR_UNSUPPORTED) return rc;
      }
      return launch_bf<TAPS, 2, WQ, 2, 1, true>(p, s);
    }
  }
  if (p.a.M % 128 == 0 && fits128) {
    if (npl == 1) return launch_bf<TAPS, 1, WQ, 4>(p, s);
    return npl == 2 ? launch_bf<TAPS, 2, WQ, 4>(p, s) : launch_bf<TAPS, 3, WQ, 4>(p, s);
  }
  if (agcn_bf16_conv_wide(TAPS, p.a.M) && (npl == 3 || npl == 1)) {   // (a window too long for the wide tile: narrow one)
    const int rc = npl == 3 ? launch_bf<TAPS, 3, 3, 2, 2>(p, s) : launch_bf<TAPS, 1, 3, 2, 2>(p, s);
    if (rc != AGCN_ERR_UNSUPPORTED) return rc;
  }
  if (npl == 1) return launch_bf<TAPS, 1, WQ, 2>(p, s);
  return npl == 2 ? launch_bf<TAPS, 2, WQ, 2>(p, s) : launch_bf<TAPS, 3, WQ, 2>(p, s);
}

}  // namespace

// 64-row problems with taps: 2 position tiles per wave (512-position workgroup tiles).  AGCN_CONV_WIDE=0 disables.
bool agcn_bf16_conv_wide(int taps, int M) {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AGCN_CONV_WIDE");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v == 1 && taps > 1 && M <= 64;
}

// internal entry points used by conv_gemm.hip's dispatch (precision: 3 = bf16x6, 2 = bf16x3)
size_t agcn_bf16_conv_workspace(int Cin, int Cout, int T, int V, int stride) {
  const int To = (T + 8 - 9) / stride + 1;
  size_t b = bf_geometry<9, 64>(V, To, stride, Cout, Cin).pack_bytes, t;
  t = bf_geometry<9, 64>(V, T, 1, Cin, Cout).pack_bytes; if (t > b) b = t;
  t = bf_geometry<9, 128>(V, To, stride, Cout, Cin).pack_bytes; if (t > b) b = t;
  t = bf_geometry<9, 128>(V, T, 1, Cin, Cout).pack_bytes; if (t > b) b = t;
  return b + 256;
}

int agcn_bf16_conv9_fwd(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* ws,
                        size_t ws_bytes, int N, int Cin, int Cout, int T, int V, int stride, int npl,
                        hipStream_t s, const float* add, int relu, const float* x_absmax) {
  BfProblem p = {};
  BfArgs& a = p.a;
  a.in = x; a.bias = bias; a.out = y; a.stats = stats_part; a.add1 = add; a.relu = relu;
  {   // AGCN_CONV_F16X3=0 keeps the forward on bf16x6
    static const int f16x3 = getenv("AGCN_CONV_F16X3") ? atoi(getenv("AGCN_CONV_F16X3")) : 1;
    p.fwd_f16 = f16x3 && npl == 3;
  }
  if (p.fwd_f16 && x_absmax) {
    a.in_absmax = x_absmax;                    // left behind by the kernel that produced x (agcn_bn_act_fwd_ex)
  } else if (p.fwd_f16) {
    // max |x| for the range scale: the last 16 bytes of the workspace (its size carries 256 bytes of slack)
    if (ws_bytes < 64) return AGCN_ERR_WORKSPACE;
    unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + ((ws_bytes - 16) & ~(size_t)15));
    if (hipMemsetAsync(amax, 0, 4, s) != hipSuccess) return AGCN_ERR_ARG;
    hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, s, x, (long)N * Cin * T * V, amax);
    if (int rc = agcn_check_launch()) return rc;
    a.in_absmax = reinterpret_cast<const float*>(amax);
    ws_bytes = (ws_bytes - 16) & ~(size_t)15;
  }
  a.N = N; a.M = Cout; a.Kinner = Cin; a.in_rows = Cin; a.V = V;
  a.T_src = T; a.T_out = (T + 8 - 9) / stride + 1; a.T_full = a.T_out;
  a.src_stride = stride; a.f_off = -4; a.out_fs = 1; a.out_fo = 0;
  p.w = w; p.sa_m = (long)Cin * 9; p.sa_c = 9; p.tap_flip_from = -1;
  p.ws = ws; p.ws_bytes = ws_bytes;
  if (stride == 1) return launch_npl<9, 2>(p, npl, s);
  return launch_npl<9, 3>(p, npl, s);
}

int agcn_bf16_conv9_bwd_data(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                             const float* mask1, const float* add2, const float* mask2, void* ws, size_t ws_bytes,
                             int N, int Cin, int Cout, int T, int V, int stride, int npl, hipStream_t s,
                             const float* dy_absmax) {
  BfProblem p = {};
  BfArgs& a = p.a;
  {   // f16x3 with the gradient normalised by its maximum (AGCN_CONV_F16X3=0: bf16x6)
    static const int f16x3 = getenv("AGCN_CONV_F16X3") ? atoi(getenv("AGCN_CONV_F16X3")) : 1;
    p.fwd_f16 = f16x3 && npl == 3;
  }
  if (p.fwd_f16 && dy_absmax) {
    a.in_absmax = dy_absmax;                   // left behind by agcn_bn_bwd_apply_ex
  } else if (p.fwd_f16) {
    if (ws_bytes < 64) return AGCN_ERR_WORKSPACE;
    unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + ((ws_bytes - 16) & ~(size_t)15));
    if (hipMemsetAsync(amax, 0, 4, s) != hipSuccess) return AGCN_ERR_ARG;
    const int To = (T + 8 - 9) / stride + 1;
    hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, s, dy, (long)N * Cout * To * V, amax);
    if (int rc = agcn_check_launch()) return rc;
    a.in_absmax = reinterpret_cast<const float*>(amax);
    ws_bytes = (ws_bytes - 16) & ~(size_t)15;
  }
  a.in = dy; a.out = dx; a.accumulate = accumulate;
  a.add1 = add1; a.mask1 = mask1; a.add2 = add2; a.mask2 = mask2;
  a.N = N; a.M = Cin; a.Kinner = Cout; a.in_rows = Cout; a.V = V;
  a.T_src = (T + 8 - 9) / stride + 1; a.T_full = T; a.src_stride = 1;
  p.w = w; p.sa_m = 9; p.sa_c = (long)Cin * 9;
  p.ws = ws; p.ws_bytes = ws_bytes;
  if (stride == 1) {
    a.T_out = T; a.out_fs = 1; a.out_fo = 0; a.f_off = -4;
    p.tap_mul = 1; p.tap_add = 0; p.tap_flip_from = 8;
    return launch_npl<9, 2>(p, npl, s);
  }
  a.T_out = (T + 1) / 2; a.out_fs = 2; a.out_fo = 0; a.f_off = -2;
  p.tap_mul = 2; p.tap_add = 0; p.tap_flip_from = 8;
  int rc = launch_npl<5, 2>(p, npl, s);
  if (rc) return rc;
  a.T_out = T / 2; a.out_fo = 1; a.f_off = -1;
  p.tap_add = 1;
  if (a.T_out > 0) rc = launch_npl<4, 2>(p, npl, s);
  return rc;
}

// 1x1 convolutions (conv_a/conv_b/down/residual; reference agcn.py:66-75,122-125) on the same kernel, TAPS = 1
int agcn_bf16_conv1_fwd(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* ws,
                        size_t ws_bytes, int N, int Cin, int Cout, int T, int V, int stride, int npl,
                        hipStream_t s, const float* add, int relu) {
  BfProblem p = {};
  BfArgs& a = p.a;
  a.in = x; a.bias = bias; a.out = y; a.stats = stats_part; a.add1 = add; a.relu = relu;
  a.N = N; a.M = Cout; a.Kinner = Cin; a.in_rows = Cin; a.V = V;
  a.T_src = T; a.T_out = (T - 1) / stride + 1; a.T_full = a.T_out;
  a.src_stride = stride; a.f_off = 0; a.out_fs = 1; a.out_fo = 0;
  p.w = w; p.sa_m = Cin; p.sa_c = 1; p.tap_flip_from = -1;
  p.ws = ws; p.ws_bytes = ws_bytes;
  if (stride == 1) return launch_npl<1, 2>(p, npl, s);
  return launch_npl<1, 3>(p, npl, s);
}

// stride 1 only (the stride-2 1x1 backward is a scatter to even frames: conv_gemm.hip keeps it)
int agcn_bf16_conv1_bwd_data(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                             const float* mask1, const float* add2, const float* mask2, void* ws, size_t ws_bytes,
                             int N, int Cin, int Cout, int T, int V, int npl, hipStream_t s) {
  BfProblem p = {};
  BfArgs& a = p.a;
  a.in = dy; a.out = dx; a.accumulate = accumulate;
  a.add1 = add1; a.mask1 = mask1; a.add2 = add2; a.mask2 = mask2;
  a.N = N; a.M = Cin; a.Kinner = Cout; a.in_rows = Cout; a.V = V;
  a.T_src = T; a.T_full = T; a.src_stride = 1;
  a.T_out = T; a.out_fs = 1; a.out_fo = 0; a.f_off = 0;
  p.w = w; p.sa_m = 1; p.sa_c = Cin; p.tap_mul = 1; p.tap_add = 0; p.tap_flip_from = 0;
  p.ws = ws; p.ws_bytes = ws_bytes;
  return launch_npl<1, 2>(p, npl, s);
}

size_t agcn_bf16_conv1_workspace(int Cin, int Cout, int T, int V, int stride) {
  const int To = (T - 1) / stride + 1;
  size_t b = bf_geometry<1, 64>(V, To, stride, Cout, Cin).pack_bytes, t;
  t = bf_geometry<1, 64>(V, T, 1, Cin, Cout).pack_bytes; if (t > b) b = t;
  t = bf_geometry<1, 128>(V, To, stride, Cout, Cin).pack_bytes; if (t > b) b = t;
  t = bf_geometry<1, 128>(V, T, 1, Cin, Cout).pack_bytes; if (t > b) b = t;
  return b + 256;
}
