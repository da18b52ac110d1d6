// Optimizer step of the training loop on ONE flat fp32 buffer: global-norm gradient clipping + Adam with L2 weight
// decay, two launches for the whole model.
//
// Reference: train.py:112-114 (torch.optim.Adam(lr, betas, weight_decay=wd), StepLR) and :63-66
// (loss.backward(); clip_grad_norm_(parameters, max_norm=20, norm_type=2); optimizer.step()).  torch walks the
// parameter list (or a multi-tensor list); here parameters, gradients and both moments live contiguously
// (mlgnn.optim.FlatAdam lays the module's parameters out that way), so the step is
//   1. adam_sumsq_kernel: per-workgroup partial sums of g^2 (fixed order -> bitwise reproducible norm),
//   2. adam_step_kernel:  every workgroup re-derives  clip = min(1, max_norm / (||g|| + 1e-6))  from the partials (no
//      host round trip, no separate scaling pass) and applies torch's single-tensor Adam formula element by element:
//        g' = clip g (+ wd p);  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;
//        p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps),   step_size = lr / (1 - b1^t)
// Parameters the backward did not reach are skipped exactly as torch skips `grad is None` (no decay, moments
// untouched): the host passes the element ranges that are live this step.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

constexpr int kAdamPartials = 256;
constexpr int kAdamMaxRanges = 64;

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
  const int64_t* ranges;        // [n_ranges][3]: first element, number of elements, elements before this range
  int n_ranges; int64_t n_live;
  const float* partial; float max_norm;
  float b1, b2, eps, wd, step_size, bias2_sqrt;
  float* norm_out;              // optional: total gradient norm (what clip_grad_norm_ returns)
};

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
  __shared__ float wsum[4];
  __shared__ int64_t rg[kAdamMaxRanges][3];
  for (int i = threadIdx.x; i < a.n_ranges * 3; i += 256) rg[i / 3][i % 3] = a.ranges[i];
  float clip = 1.f;
  if (a.max_norm > 0.f) {
    float acc = threadIdx.x < kAdamPartials ? a.partial[threadIdx.x] : 0.f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float norm = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    clip = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
    if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) a.norm_out[0] = norm;
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_live; i += (int64_t)gridDim.x * 256) {
    int r = 0;
    while (r + 1 < a.n_ranges && i >= rg[r + 1][2]) ++r;
    const int64_t e = rg[r][0] + (i - rg[r][2]);
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
                               const int64_t* live_ranges, int n_ranges, int64_t n_live, float max_norm, float beta1,
                               float beta2, float eps, float weight_decay, float step_size, float bias2_sqrt,
                               float* workspace, void* stream) {
  if (n < 0 || n_live < 0 || n_live > n || n_ranges < 0 || n_ranges > kAdamMaxRanges) return MLGNN_E_SHAPE;
  if (n == 0 || n_live == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !live_ranges || !workspace) return MLGNN_E_NULL;
  if (!(bias2_sqrt > 0.f)) return MLGNN_E_MODE;
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.f)
    hipLaunchKernelGGL(adam_sumsq_kernel, dim3(kAdamPartials), dim3(256), 0, s, grads, n, workspace);
  AdamArgs a{params, grads, exp_avg, exp_avg_sq, live_ranges, n_ranges, n_live, workspace, max_norm,
             beta1, beta2, eps, weight_decay, step_size, bias2_sqrt, workspace + kAdamPartials};
  int64_t blocks = (n_live + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
