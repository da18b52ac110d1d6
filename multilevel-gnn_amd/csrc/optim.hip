// Optimizer step of the training loop on ONE flat fp32 buffer: global-norm gradient clipping + Adam with L2 weight
// decay, two launches for the whole model.
//
// Reference: train.py:112-114 (torch.optim.Adam(lr, betas, weight_decay=wd), StepLR) and :63-66
// (loss.backward(); clip_grad_norm_(parameters, max_norm=20, norm_type=2); optimizer.step()).  torch walks the
// parameter list (or a multi-tensor list); here parameters, gradients and both moments live contiguously
// (mlgnn.optim.FlatAdam lays the module's parameters out that way), so the step is
//   1. adam_sumsq_kernel: per-workgroup partial sums of g^2 (fixed order -> bitwise reproducible norm),
//   2. adam_step_kernel:  every workgroup re-derives  clip = min(1, max_norm / (||g|| + 1e-6))  from the partials (no
//      host round trip, no separate scaling pass; a NaN norm gives a NaN factor, as clip_grad_norm_ does) and applies
//      torch's single-tensor Adam formula element by element:
//        g' = clip g (+ wd p);  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;
//        p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps),   step_size = lr / (1 - b1^t)
// Parameters the backward did not reach are skipped exactly as torch skips `grad is None` (no decay, moments
// untouched).  Which parameters are live is DEVICE data -- one float per parameter, > 0 = live -- so that under data
// parallelism the flags can ride at the tail of the gradient all-reduce and every rank steps the union of what any
// rank reached (mlgnn/dist.py); an element finds its parameter by bisection over the parameter offsets.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

constexpr int kAdamPartials = 256;
constexpr int kAdamLdsParams = 2048;          // parameter offsets kept in LDS up to this many parameters

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float wsum[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

struct AdamArgs {
  float* p; float* g; float* m; float* v;
  int64_t n;
  const int64_t* offsets;       // [n_params + 1]: first element of every parameter, offsets[n_params] = n
  const float* live;            // [n_params]: > 0 = the parameter received a gradient; nullptr = all live
  int n_params;
  const float* partial; float max_norm;
  float b1, b2, eps, wd, step_size, bias2_sqrt;
  float* norm_out;              // optional: total gradient norm (what clip_grad_norm_ returns)
};

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
  __shared__ float wsum[4];
  __shared__ int64_t off_lds[kAdamLdsParams + 1];
  const bool in_lds = a.n_params <= kAdamLdsParams;
  if (in_lds)
    for (int i = threadIdx.x; i <= a.n_params; i += 256) off_lds[i] = a.offsets[i];
  float clip = 1.f;
  if (a.max_norm > 0.f) {
    float acc = threadIdx.x < kAdamPartials ? a.partial[threadIdx.x] : 0.f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float norm = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    // clip_grad_norm_: coef = max_norm / (norm + 1e-6) clamped to <= 1; a NaN norm makes the coefficient NaN and
    // poisons every gradient (fminf alone would return 1 and hide it)
    clip = (norm == norm) ? fminf(a.max_norm / (norm + 1e-6f), 1.0f) : norm;
    if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) a.norm_out[0] = norm;
  }
  __syncthreads();
  const int64_t* off = in_lds ? off_lds : a.offsets;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (int64_t)gridDim.x * 256) {
    if (a.live) {
      int lo = 0, hi = a.n_params;                       // largest lo with off[lo] <= e  (empty parameters repeat an
      while (hi - lo > 1) {                              //  offset: the bisection lands on the last of them, the one
        const int mid = (lo + hi) >> 1;                  //  that owns the element)
        if (off[mid] <= e) lo = mid; else hi = mid;
      }
      if (!(a.live[lo] > 0.f)) continue;
    }
    const float p = a.p[e];
    float g = a.g[e] * clip;
    if (a.max_norm > 0.f) a.g[e] = g;                    // clip_grad_norm_ scales the gradients in place
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);
    const float m0 = a.m[e];
    const float m = m0 + (1.f - a.b1) * (g - m0);
    const float v = a.v[e] * a.b2 + (1.f - a.b2) * g * g;
    a.m[e] = m;
    a.v[e] = v;
    a.p[e] = p - a.step_size * (m / (sqrtf(v) / a.bias2_sqrt + a.eps));
  }
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_adam_workspace_floats(void) { return kAdamPartials + 1; }

extern "C" int mlgnn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                               const int64_t* param_offsets, const float* live, int64_t n_params, float max_norm,
                               float beta1, float beta2, float eps, float weight_decay, float step_size,
                               float bias2_sqrt, float* workspace, void* stream) {
  if (n < 0 || n_params < 0 || n_params >= (1ll << 31)) return MLGNN_E_SHAPE;
  if (n == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !workspace) return MLGNN_E_NULL;
  if (live && (!param_offsets || n_params == 0)) return MLGNN_E_NULL;
  if (!(bias2_sqrt > 0.f)) return MLGNN_E_MODE;
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.f)
    hipLaunchKernelGGL(adam_sumsq_kernel, dim3(kAdamPartials), dim3(256), 0, s, grads, n, workspace);
  AdamArgs a{params, grads, exp_avg, exp_avg_sq, n, param_offsets, live, (int)n_params, workspace, max_norm,
             beta1, beta2, eps, weight_decay, step_size, bias2_sqrt, workspace + kAdamPartials};
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
