// Backward prologue of the `power` aggregator (reference: models/gcn_lib/sparse/torch_message.py:66-76,
//     out = clamp(mean_e clamp(m_e, 1e-7, 10)^p, 1e-7, 10)^(1/p) ):
// the aggregation backward (csrc/aggregate_bwd.hip) takes the cotangent already carried through the OUTER power and the
// mean,   q = grad_out * mu_c^(1/p - 1) * [1e-7 <= mu <= 10] / max(deg, 1),   mu = the forward's mean (aux), mu_c its clamp,
// and a learnable p also needs   d loss / dp = sum grad_out * out * (-ln(mu_c) / p^2 + [in range] * aux2 / (p * mu_c)).
// One streaming pass instead of the six elementwise ATen passes (+ two reductions) this used to be.  Built with default
// NaN semantics (the clamp carries a NaN like torch.clamp; the aggregation translation units are not).
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

constexpr float kPwLo = 1e-7f, kPwHi = 1e1f;      // torch_message.py:69
constexpr int kPwBlocks = 1024;

struct PowArgs {
  const void* go; const float* mu; const int* rowptr; const void* out; const float* aux2; const float* p_dev; float p;
  void* q; float* partial; int64_t n; int d; int bf16; int learn_p;
};

__global__ __launch_bounds__(kBlock) void power_bwd_prologue_kernel(const PowArgs a) {
  __shared__ float wsum[kWavesPerBlock];
  const float p = a.p_dev ? a.p_dev[0] : a.p;
  const float e = 1.0f / p - 1.0f;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
    const int row = (int)(i / a.d);
    const float g = a.bf16 ? bf16_to_f32(static_cast<const uint16_t*>(a.go)[i]) : static_cast<const float*>(a.go)[i];
    const float mu = a.mu[i];
    const float mc = (mu != mu) ? mu : fminf(fmaxf(mu, kPwLo), kPwHi);
    const float inr = (mu >= kPwLo && mu <= kPwHi) ? 1.0f : 0.0f;
    const float deg = (float)max(a.rowptr[row + 1] - a.rowptr[row], 1);
    const float q = g * powf(mc, e) * inr / deg;
    if (a.bf16) static_cast<uint16_t*>(a.q)[i] = f32_to_bf16(q);
    else static_cast<float*>(a.q)[i] = q;
    if (a.learn_p) {
      const float o = a.bf16 ? bf16_to_f32(static_cast<const uint16_t*>(a.out)[i]) : static_cast<const float*>(a.out)[i];
      acc += g * o * (-logf(mc) / (p * p) + inr * a.aux2[i] / (p * mc));
    }
  }
  if (a.learn_p) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kWavesPerBlock; ++w) t += wsum[w];
      a.partial[blockIdx.x] = t;
    }
  }
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_power_bwd_prologue_workspace_floats(void) { return kPwBlocks; }

extern "C" int mlgnn_power_bwd_prologue(const void* grad_out, const float* mu, const int32_t* rowptr, const void* out,
                                        const float* aux2, float p, const float* p_dev, void* q, float* grad_p,
                                        float* workspace, int64_t workspace_floats, int64_t N, int64_t d, int dtype,
                                        void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (N < 0 || d <= 0 || N > INT32_MAX) return MLGNN_E_SHAPE;
  if (N == 0) return grad_p ? (int)hipMemsetAsync(grad_p, 0, 4, (hipStream_t)stream) : 0;
  if (!grad_out || !mu || !rowptr || !q) return MLGNN_E_NULL;
  if (!p_dev && !(p != 0.0f)) return MLGNN_E_MODE;
  const bool learn = grad_p != nullptr;
  if (learn && (!out || !aux2 || !workspace)) return MLGNN_E_NULL;
  if (learn && workspace_floats < kPwBlocks) return MLGNN_E_WORKSPACE;
  PowArgs a;
  a.go = grad_out; a.mu = mu; a.rowptr = rowptr; a.out = out; a.aux2 = aux2; a.p_dev = p_dev; a.p = p;
  a.q = q; a.partial = workspace; a.n = N * d; a.d = (int)d; a.bf16 = dtype == MLGNN_DTYPE_BF16; a.learn_p = learn;
  int64_t blocks = (a.n + kBlock - 1) / kBlock;
  if (blocks > kPwBlocks) blocks = kPwBlocks;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(power_bwd_prologue_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  int err = (int)hipGetLastError();
  if (err || !learn) return err;
  launch_reduce_partials(workspace, grad_p, (int)blocks, 1, s);
  return (int)hipGetLastError();
}
