// Common device/host helpers for the agcn_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AGCN_OK 0
#define AGCN_ERR_ARG (-1)
#define AGCN_ERR_WORKSPACE (-2)
#define AGCN_ERR_UNSUPPORTED (-3)

// exact-f32 matrix core op: D(32x32) += A(32x2) * B(2x32); lane l holds A[l&31][l>>5], B[l>>5][l&31];
// D register j of lane l is D[row = (j&3) + 8*(j>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

// ReLU masks reach the kernels either as the fp32 activation tensor (positive = pass) or as its sign bit mask
// (agcn_bn_act_fwd: bit e of word w <-> element 32*w + e); the raw 32-bit load of either kind and the test:
__device__ __forceinline__ float mask_load(const float* m, long idx, int bits) {
  return bits ? m[idx >> 5] : m[idx];            // bit-mask words are fetched as raw 32-bit patterns
}
__device__ __forceinline__ bool mask_pass(float raw, long idx, int bits) {
  return bits ? ((__builtin_bit_cast(unsigned, raw) >> (idx & 31)) & 1u) != 0u : raw > 0.f;
}

// sum over the 32 lanes of a half-wave (lanes l and l^k stay inside the half for k<32)
__device__ __forceinline__ float half_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// Bijective XCD-aware remap of a 1-D block id (guide T1): blocks b and b+8 share an XCD, so give
// every XCD a contiguous range of logical ids (neighbouring tiles share halos / rows in its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, k = bid >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + k;
}

extern "C" size_t agcn_colsum_scratch_bytes(int W);
extern "C" int agcn_colsum(const float* X, int nslots, int W, void* scratch, float* out, void* stream);

// adjacency.hip: slab sum + 1/K + column softmax + graph terms (also the tail of adj_fused.hip's forward)
int agcn_adj_finalize(const float* spart, const float* A, const float* PA, const float* alpha, float* P, float* adj,
                      int N, int Ci, int T, int V, hipStream_t s, int nused = -1);

// persistent weight-stationary forward of the adjacency scores (adj_ws.hip)
bool agcn_adj_ws_supported(int N, int C, int Ci, int T, int V);
size_t agcn_adj_ws_workspace(int C, int Ci);
int agcn_adj_ws_scores(const float* x, const float* wab, const float* bab, float* tp_out, float* spart, int slots,
                       int* nslots_used, const float* x_absmax, void* ws, size_t ws_bytes, int N, int C, int Ci, int T, int V,
                       hipStream_t s);

// split-bf16 temporal convolution (conv_gemm_bf16.hip); npl: 3 = bf16x6 (fp32-equivalent), 2 = bf16x3
size_t agcn_bf16_conv_workspace(int Cin, int Cout, int T, int V, int stride);
bool agcn_bf16_conv_wide(int taps, int M);
int agcn_bf16_conv9_fwd(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* ws,
                        size_t ws_bytes, int N, int Cin, int Cout, int T, int V, int stride, int npl, hipStream_t s,
                        const float* add = nullptr, int relu = 0, const float* x_absmax = nullptr);
int agcn_bf16_conv9_bwd_data(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                             const float* mask1, const float* add2, const float* mask2, void* ws, size_t ws_bytes,
                             int N, int Cin, int Cout, int T, int V, int stride, int npl, hipStream_t s,
                             const float* dy_absmax = nullptr);

size_t agcn_bf16_conv1_workspace(int Cin, int Cout, int T, int V, int stride);
int agcn_bf16_conv1_fwd(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* ws,
                        size_t ws_bytes, int N, int Cin, int Cout, int T, int V, int stride, int npl, hipStream_t s,
                        const float* add = nullptr, int relu = 0);
int agcn_bf16_conv1_bwd_data(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                             const float* mask1, const float* add2, const float* mask2, void* ws, size_t ws_bytes,
                             int N, int Cin, int Cout, int T, int V, int npl, hipStream_t s);

// register-chained aggregate+project (gcn_chain.hip); mode 0 = forward, 1 = backward-data
bool agcn_gcn_chain_supported(int M, int K, int V);
int agcn_chain_f16x3();                       // 1: the chain runs on f16x3 (split_f16.h), 0: bf16x6 / AGCN_GEMM's mode
int agcn_gcn_chain_tiles(int T);
int agcn_gcn_chain_stats_slots(int N, int M, int K, int T, int V);
size_t agcn_gcn_chain_workspace(int M, int K, int K2, int T, int V);
int agcn_gcn_chain(int mode, const float* in, const float* adj, const float* wcat, const float* bias, float* out,
                   float* stats_part, int accumulate, const float* add1, const float* mask1, const float* add2,
                   const float* mask2, int mask_bits, const float* in2, const float* w2, int K2, void* ws, size_t ws_bytes,
                   int N, int C, int Cout, int T, int V, hipStream_t stream, int relu = 0, int w2_rows_are_outputs = 0,
                   const float* in_absmax = nullptr, const float* in2_absmax = nullptr);

bool agcn_gcn_dadj_chain_supported(int C, int V);
int agcn_gcn_dadj_chain_slots(int C, int T);
size_t agcn_gcn_dadj_chain_workspace(int C, int Cout);
int agcn_gcn_dadj_chain(const float* dy, const float* wcat, const float* x, float* dadj_part, void* ws, size_t ws_bytes,
                        int N, int C, int Cout, int T, int V, hipStream_t stream, const float* dy_absmax = nullptr,
                        const float* x_absmax = nullptr);

// split-bf16 weight gradients of the tap-free contractions (wgrad_chain.hip): partial slabs only, reduced by the caller
bool agcn_wgrad_chain_supported(int M, int C, int V);
size_t agcn_wgrad_chain_workspace(int agg, int N, int M, int C, int V, int T_out);
int agcn_wgrad_chain(int agg, const float* dy, const float* x, const float* adj, void* ws, size_t ws_bytes, int* nslabs,
                     int N, int M, int C, int V, int T_src, int T_out, int stride, hipStream_t s,
                     const float* dy_absmax = nullptr, const float* x_absmax = nullptr);

// split-bf16 weight gradient of the 9-tap temporal convolution (wgrad9_bf16.hip): slabs [nslabs][9][M][C] at ws
bool agcn_wgrad9_bf16_supported(int M, int C, int V, int stride);
size_t agcn_wgrad9_bf16_workspace(int N, int M, int C, int V, int T, int stride);
int agcn_wgrad9_bf16(const float* dy, const float* x, void* ws, size_t ws_bytes, int* nslabs, int N, int M, int C, int V,
                     int T, int stride, hipStream_t s, const float* dy_absmax = nullptr, const float* x_absmax = nullptr);

// GEMM arithmetic of the channel contractions: 3 = bf16x6 (default: fp32-equivalent accuracy, measured), 0 = f32 MFMA,
// 2 = bf16x3 (~5e-6 per GEMM; does NOT hold the 1e-4 parity bar end to end), 1 = bf16 (plain bf16 MFMA operands, ONE
// product per fp32 product, fp32 accumulate and fp32 storage: BASELINE configs[3], ~2^-9 relative per product).
// Chosen once per process from the environment variable AGCN_GEMM (bf16x6 | f32 | bf16x3 | bf16).
static inline int agcn_gemm_precision() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("AGCN_GEMM");
    mode = 3;
    if (e && !strcmp(e, "f32")) mode = 0;
    else if (e && !strcmp(e, "bf16x3")) mode = 2;
    else if (e && !strcmp(e, "bf16")) mode = 1;
  }
  return mode;
}
// the register-chained / split-plane kernel family serves bf16x6 (3 planes, 6 products) and bf16 (hi plane, 1 product)
static inline bool agcn_chained() { const int m = agcn_gemm_precision(); return m == 3 || m == 1; }
static inline int agcn_npl() { return agcn_gemm_precision() == 1 ? 1 : 3; }

// Raise a kernel's dynamic-LDS limit to 160 KB on the CURRENT device, once per (kernel, device).  hipFuncSetAttribute
// is per device, and forward / autograd-backward threads may reach a first launch together: the flags are atomics and
// a lost race only repeats the idempotent call.  `done` is a static array owned by the launcher instantiation.
#define AGCN_MAX_DEVICES 64
static inline int agcn_allow_big_lds_rt(const void* kern, unsigned char* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= AGCN_MAX_DEVICES) dev = -1;
  if (dev >= 0 && __atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) return AGCN_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0) __atomic_store_n(&done[dev], (unsigned char)1, __ATOMIC_RELEASE);
  return AGCN_OK;
}

// Diagnostic only: the contraction launchers note which kernel instantiation they enqueued last ON THIS THREAD, so that
// a benchmark can name the kernel it actually timed (agcn_last_kernel()).  Never read by any compute path.
#include <stdio.h>
inline thread_local char agcn_last_kernel_buf[192] = "";
#define AGCN_NOTE_KERNEL(...) snprintf(agcn_last_kernel_buf, sizeof(agcn_last_kernel_buf), __VA_ARGS__)

static inline int agcn_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AGCN_OK : (int)e;
}
