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

// 16-byte loads, four of them in flight per thread (the scalar one-load-per-iteration form ran at 0.9 TB/s on the 175 MB
// gradient of the kirc-shape head: 197 us).  The summation order is a function of (n, grid) only: bitwise reproducible.
__global__ __launch_bounds__(256) void adam_sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float wsum[4];
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const int64_t units = n / 4, stride = (int64_t)gridDim.x * 256;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < units; i += 4 * stride) {
    const float4 q0 = g4[i], q1 = g4[i + stride], q2 = g4[i + 2 * stride], q3 = g4[i + 3 * stride];
    a0 = fmaf(q0.x, q0.x, fmaf(q0.y, q0.y, fmaf(q0.z, q0.z, fmaf(q0.w, q0.w, a0))));
    a1 = fmaf(q1.x, q1.x, fmaf(q1.y, q1.y, fmaf(q1.z, q1.z, fmaf(q1.w, q1.w, a1))));
    a2 = fmaf(q2.x, q2.x, fmaf(q2.y, q2.y, fmaf(q2.z, q2.z, fmaf(q2.w, q2.w, a2))));
    a3 = fmaf(q3.x, q3.x, fmaf(q3.y, q3.y, fmaf(q3.z, q3.z, fmaf(q3.w, q3.w, a3))));
  }
  for (; i < units; i += stride) {
    const float4 q = g4[i];
    a0 = fmaf(q.x, q.x, fmaf(q.y, q.y, fmaf(q.z, q.z, fmaf(q.w, q.w, a0))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - units * 4)) {            // the last n % 4 elements
    const float t = g[units * 4 + threadIdx.x];
    a1 = fmaf(t, t, a1);
  }
  float acc = (a0 + a1) + (a2 + a3);
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
  auto owner = [&](int64_t e) {                          // largest lo with off[lo] <= e  (empty parameters repeat an
    int lo = 0, hi = a.n_params;                         //  offset: the bisection lands on the last of them, the one
    while (hi - lo > 1) {                                //  that owns the element)
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
  };
  auto update = [&](float p, float g0, float m0, float v0, float& g_out, float& m_out, float& v_out) {
    float g = g0 * clip;
    g_out = g;
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);
    const float m = m0 + (1.f - a.b1) * (g - m0);
    const float v = v0 * a.b2 + (1.f - a.b2) * g * g;
    m_out = m;
    v_out = v;
    return p - a.step_size * (m / (sqrtf(v) / a.bias2_sqrt + a.eps));
  };
  auto one = [&](int64_t e) {
    if (a.live && !(a.live[owner(e)] > 0.f)) return;
    float g, m, v;
    const float p = update(a.p[e], a.g[e], a.m[e], a.v[e], g, m, v);
    if (a.max_norm > 0.f) a.g[e] = g;                    // clip_grad_norm_ scales the gradients in place
    a.m[e] = m; a.v[e] = v; a.p[e] = p;
  };
  // 16 bytes of each of p, g, m, v per thread and step.  A unit that lies inside one parameter (always, with the 16-byte
  // aligned slots of mlgnn.optim.FlatAdam) is looked up once; the owner found last is re-used while the walk stays inside it.
  const int64_t units = a.n / 4;
  float4* p4 = reinterpret_cast<float4*>(a.p);
  float4* g4 = reinterpret_cast<float4*>(a.g);
  float4* m4 = reinterpret_cast<float4*>(a.m);
  float4* v4 = reinterpret_cast<float4*>(a.v);
  int64_t own_lo = 0, own_hi = -1;                       // element range of the cached owner
  bool own_live = true;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
    const int64_t e = u * 4;
    if (a.live) {
      if (!(e >= own_lo && e + 3 < own_hi)) {
        const int lo = owner(e);
        own_lo = off[lo]; own_hi = off[lo + 1]; own_live = a.live[lo] > 0.f;
        if (e + 3 >= own_hi) {                           // the unit straddles parameters: element by element
          for (int j = 0; j < 4; ++j) one(e + j);
          own_hi = -1;
          continue;
        }
      }
      if (!own_live) continue;
    }
    const float4 p = p4[u], g = g4[u], m = m4[u], v = v4[u];
    float4 po, go, mo, vo;
    po.x = update(p.x, g.x, m.x, v.x, go.x, mo.x, vo.x);
    po.y = update(p.y, g.y, m.y, v.y, go.y, mo.y, vo.y);
    po.z = update(p.z, g.z, m.z, v.z, go.z, mo.z, vo.z);
    po.w = update(p.w, g.w, m.w, v.w, go.w, mo.w, vo.w);
    if (a.max_norm > 0.f) g4[u] = go;
    m4[u] = mo; v4[u] = vo; p4[u] = po;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(a.n - units * 4)) one(units * 4 + threadIdx.x);
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
  if (((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
        reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) != 0)
    return MLGNN_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.f)
    hipLaunchKernelGGL(adam_sumsq_kernel, dim3(kAdamPartials), dim3(256), 0, s, grads, n, workspace);
  AdamArgs a{params, grads, exp_avg, exp_avg_sq, n, param_offsets, live, (int)n_params, workspace, max_norm,
             beta1, beta2, eps, weight_decay, step_size, bias2_sqrt, workspace + kAdamPartials};
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
