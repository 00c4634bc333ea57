// Loss heads and the optimizer step.
//   gcmi_loss_fwd_bwd: SoftmaxCrossEntropy (models/losses.py:251-259) or L2Loss
//     (:85-94) through _StandardLoss (models/torch_models/torch_model.py:1275-1294):
//     loss = mean over (rows, tasks) of w*l, plus d loss / d logits in the same pass.
//   gcmi_adam_step: torch.optim.Adam as models/optimizers.py:231-241 configures it.
// All tiny and launch-bound; one thread per (row, task) / per parameter.
#include <math.h>

#include <algorithm>

#include "common.h"

namespace gcmi {

constexpr int kLBlock = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < kLBlock / 64; ++w) t += red[w];
  return t;  // valid on thread 0
}

__global__ void __launch_bounds__(kLBlock)
loss_kernel(int kind, const float* __restrict__ logits, const float* __restrict__ labels,
            const float* __restrict__ weights, int64_t n_items, int n_classes, float inv_count,
            float* __restrict__ dlogits, float* __restrict__ probs, double* __restrict__ acc) {
  __shared__ double red[kLBlock / 64];
  double local = 0.0;
  for (int64_t it = (int64_t)blockIdx.x * kLBlock + threadIdx.x; it < n_items;
       it += (int64_t)gridDim.x * kLBlock) {
    const float w = weights ? weights[it] : 1.f;
    if (kind == 0) {
      const float* x = logits + it * n_classes;
      const float* y = labels + it * n_classes;
      float m = -INFINITY;
      for (int c = 0; c < n_classes; ++c) m = fmaxf(m, x[c]);
      float se = 0.f, ysum = 0.f;
      for (int c = 0; c < n_classes; ++c) {
        se += expf(x[c] - m);
        ysum += y[c];
      }
      const float lse = logf(se);
      float l = 0.f;
      for (int c = 0; c < n_classes; ++c) {
        const float logp = x[c] - m - lse;
        const float p = expf(logp);
        l -= y[c] * logp;
        if (dlogits) dlogits[it * n_classes + c] = w * (p * ysum - y[c]) * inv_count;
        if (probs) probs[it * n_classes + c] = p;
      }
      local += (double)(w * l);
    } else {
      const float dlt = logits[it] - labels[it];
      local += (double)(w * dlt * dlt);
      if (dlogits) dlogits[it] = 2.f * dlt * w * inv_count;
    }
  }
  const double t = block_sum(local, red);
  if (threadIdx.x == 0) atomicAdd(acc, t);
}

__global__ void loss_finalize_kernel(double* __restrict__ acc, float inv_count,
                                     float* __restrict__ loss, int n_rep) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < n_rep; ++i) {
      t += acc[i];
      acc[i] = 0.0;  // a scratch that starts clean stays clean
    }
    *loss = (float)(t * (double)inv_count);
  }
}

__global__ void __launch_bounds__(kLBlock)
softmax_kernel(const float* __restrict__ logits, int64_t n_items, int n_classes,
               float* __restrict__ probs) {
  for (int64_t it = (int64_t)blockIdx.x * kLBlock + threadIdx.x; it < n_items;
       it += (int64_t)gridDim.x * kLBlock) {
    const float* x = logits + it * n_classes;
    float m = -INFINITY;
    for (int c = 0; c < n_classes; ++c) m = fmaxf(m, x[c]);
    float se = 0.f;
    for (int c = 0; c < n_classes; ++c) se += expf(x[c] - m);
    const float inv = 1.f / se;
    for (int c = 0; c < n_classes; ++c) probs[it * n_classes + c] = expf(x[c] - m) * inv;
  }
}

__global__ void __launch_bounds__(kLBlock)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, float one_minus_b1, float b2, float one_minus_b2,
            float step_size, float inv_bc2_sqrt, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * kLBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kLBlock) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * one_minus_b1;          // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + gi * gi * one_minus_b2;         // mul_(beta2).addcmul_(g, g, 1-beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;          // sqrt(v)/sqrt(bc2) + eps
    p[i] -= step_size * (mi / denom);                            // addcdiv_(m, denom, -lr/bc1)
  }
}

int loss_impl(int32_t kind, const float* d_logits, const float* d_labels, const float* d_weights,
              int64_t n_rows, int32_t n_tasks, int32_t n_classes, float* d_loss, float* d_dlogits,
              float* d_probs, double* d_acc, bool acc_clean, void* stream) {
  GCMI_CHECK_ARG(kind == 0 || kind == 1, "loss: kind must be 0 (softmax CE) or 1 (L2)");
  GCMI_CHECK_ARG(n_rows > 0 && n_tasks > 0 && (kind == 1 || n_classes > 0), "loss: bad shape");
  GCMI_CHECK_ARG(d_logits && d_labels && d_loss && d_acc, "loss: NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  if (!acc_clean && hipMemsetAsync(d_acc, 0, sizeof(double), st) != hipSuccess) {
    set_error("loss: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  const int64_t n_items = n_rows * n_tasks;
  const float inv_count = 1.f / (float)n_items;
  // every workgroup ends with ONE fp64 atomic on the same address: a few hundred workgroups with a grid-stride
  // loop, not one per 256 items (3 072 same-address atomics took most of the kernel's 43 us)
  const int loss_blocks = std::min(grid_for(n_items, kLBlock), 256);
  hipLaunchKernelGGL(loss_kernel, dim3(loss_blocks), dim3(kLBlock), 0, st, kind,
                     d_logits, d_labels, d_weights, n_items, kind == 0 ? n_classes : 1, inv_count,
                     d_dlogits, kind == 0 ? d_probs : nullptr, d_acc);
  GCMI_CHECK_LAUNCH("loss");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, d_acc, inv_count, d_loss, 1);
  GCMI_CHECK_LAUNCH("loss_finalize");
  return GCMI_OK;
}

// *d_loss = (sum of the n_rep accumulator replicas) * inv_count, and the accumulators are left clean (head_bwd.hip adds
// to them itself)
int loss_finalize_impl(double* d_acc, float inv_count, float* d_loss, void* stream, int n_rep) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_acc, inv_count, d_loss, n_rep);
  GCMI_CHECK_LAUNCH("loss_finalize");
  return GCMI_OK;
}

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int gcmi_loss_fwd_bwd(int32_t kind, const float* d_logits, const float* d_labels,
                      const float* d_weights, int64_t n_rows, int32_t n_tasks,
                      int32_t n_classes, float* d_loss, float* d_dlogits, float* d_probs,
                      double* d_acc, void* stream) {
  return loss_impl(kind, d_logits, d_labels, d_weights, n_rows, n_tasks, n_classes, d_loss, d_dlogits,
                   d_probs, d_acc, false, stream);
}

int gcmi_softmax(const float* d_logits, int64_t n_rows_tasks, int32_t n_classes, float* d_probs,
                 void* stream) {
  GCMI_CHECK_ARG(n_rows_tasks >= 0 && n_classes > 0, "softmax: bad shape");
  if (n_rows_tasks == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_logits && d_probs, "softmax: NULL buffer");
  hipLaunchKernelGGL(softmax_kernel, dim3(grid_for(n_rows_tasks, kLBlock)), dim3(kLBlock), 0,
                     (hipStream_t)stream, d_logits, n_rows_tasks, n_classes, d_probs);
  GCMI_CHECK_LAUNCH("softmax");
  return GCMI_OK;
}

int gcmi_adam_step(float* d_param, const float* d_grad, float* d_m, float* d_v, int64_t n,
                   float lr, float beta1, float beta2, float eps, int64_t step, void* stream) {
  GCMI_CHECK_ARG(n >= 0 && step >= 1, "adam: bad n/step");
  if (n == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_param && d_grad && d_m && d_v, "adam: NULL buffer");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, kLBlock)), dim3(kLBlock), 0, (hipStream_t)stream,
                     d_param, d_grad, d_m, d_v, n, 1.f - beta1, beta2, 1.f - beta2, step_size,
                     inv_bc2_sqrt, eps);
  GCMI_CHECK_LAUNCH("adam");
  return GCMI_OK;
}

}  // extern "C"
