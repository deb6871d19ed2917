// Measured fp32 matrix-core ceiling on this device: back-to-back v_mfma_f32_32x32x2_f32, operands in registers
// (variant 0) or re-read from LDS every step (variant 1), random data, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LDSREAD>
__global__ void __launch_bounds__(256) k(const float* in, float* out, int iters) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x16 acc[4];
  for (int z = 0; z < 4; ++z) for (int j = 0; j < 16; ++j) acc[z][j] = 0.f;
  float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768];
  int off = tid & 63;
  for (int it = 0; it < iters; ++it) {
    if (LDSREAD) {
      a0 = lds[off]; a1 = lds[off + 64]; b0 = lds[off + 128]; b1 = lds[off + 192];
      off = (off + 256) & 4095;
    }
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
  }
  float s = 0;
  for (int z = 0; z < 4; ++z) for (int j = 0; j < 16; ++j) s += acc[z][j];
  out[blockIdx.x * 256 + tid] = s;
}
int main() {
  float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 256 * 4);
  std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 2; ++variant)
    for (int wgs : {256, 512, 1024}) {
      const int iters = 20000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)wgs * 4 * iters * 4 * 4096.0;
        if (rep) printf("variant %d (lds %d) wgs %d: %.3f ms  %.1f TFLOP/s\n", variant, variant, wgs, ms, flops / ms / 1e9);
      }
    }
  return 0;
}
