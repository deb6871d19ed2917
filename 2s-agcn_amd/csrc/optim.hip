// Training-step tail of the reference Processor loop (utils/processor.py:697-703) on ONE flat fp32 buffer:
// global grad-norm clip (torch.nn.utils.clip_grad_norm_, max_norm, eps 1e-6) fused with the SGD update
// (momentum, optional Nesterov, L2 weight decay; torch.optim.SGD semantics, dampening 0).
#include "agcn_common.h"

namespace {

__global__ void __launch_bounds__(256)
sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = g[i];
    s += v * v;
  }
  s = half_sum(s);
  s += __shfl_xor(s, 32);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm_out[0] = grad_scale * sqrt(sum) ; norm_out[1] = clip coefficient applied to grad_scale*grad
__global__ void norm_finalize_kernel(const float* __restrict__ part, int nparts, float grad_scale, float max_norm,
                                     float* __restrict__ norm_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += (double)part[i];
  const float norm = grad_scale * (float)sqrt(s);
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  norm_out[0] = norm;
  norm_out[1] = coef;
}

__global__ void __launch_bounds__(256)
sgd_update_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr,
                  float momentum, float wd, int nesterov, float grad_scale, int first_step,
                  const float* __restrict__ norm) {
  const float gs = grad_scale * norm[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pv = p[i];
    float d = g[i] * gs + wd * pv;
    if (momentum != 0.f) {
      const float b = first_step ? d : momentum * buf[i] + d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] = pv - lr * d;
  }
}

constexpr int SGD_BLOCKS = 1024;

}  // namespace

extern "C" {

size_t agcn_sgd_step_workspace(long n) { (void)n; return sizeof(float) * (SGD_BLOCKS + 2); }

// param/grad/momentum_buf: n floats.  grad_scale multiplies the gradient first (1/world_size after a SUM all-reduce).
// norm_out: 2 device floats {total grad norm (after grad_scale), clip coefficient}.  max_norm <= 0 disables clipping.
int agcn_sgd_step(float* param, const float* grad, float* momentum_buf, long n, float lr, float momentum,
                  float weight_decay, int nesterov, float max_norm, float grad_scale, int first_step, void* workspace,
                  size_t workspace_bytes, float* norm_out, void* stream) {
  if (!param || !grad || !momentum_buf || !workspace || !norm_out || n <= 0) return AGCN_ERR_ARG;
  if (workspace_bytes < sizeof(float) * (SGD_BLOCKS + 2)) return AGCN_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SGD_BLOCKS), dim3(256), 0, s, grad, n, part);
  int rc = agcn_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)part, SGD_BLOCKS, grad_scale,
                     max_norm, norm_out);
  rc = agcn_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(sgd_update_kernel, dim3(SGD_BLOCKS), dim3(256), 0, s, param, grad, momentum_buf, n, lr, momentum,
                     weight_decay, nesterov, grad_scale, first_step, (const float*)norm_out);
  return agcn_check_launch();
}

}  // extern "C"
