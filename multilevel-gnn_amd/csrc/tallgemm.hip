// Tall-skinny fp32 GEMM on the bf16 matrix cores with 3-way split precision:
//     C[N,J] = A[N,R] * Bt[J,R]^T (+ bias[J]),     N >> R, J  (R, J <= 256)
//
// Reference: the nn.Linear layers of MLP (models/gcn_lib/sparse/torch_nn.py:54-75) applied to every
// node row -- forward (A = activations, Bt = weight) and input gradient (A = grad_out, Bt = weight^T).
// The fp32 MFMA runs at 1/16 of the bf16 rate, and a 640 000 x 128 x 256 product is MFMA-bound on it
// (~0.4 ms on the library); in bf16 the same product is HBM-bound.  Each fp32 operand is split as
//   a = a_hi + a_lo,  a_hi = bf16(a),  a_lo = bf16(a - a_hi)        (|a - a_hi - a_lo| <= 2^-18 |a|)
// and the product is accumulated in fp32 from three MFMAs: a_hi b_hi + a_lo b_hi + a_hi b_lo
// (dropped: a_lo b_lo <= 2^-18 |ab|).  Worst-case relative error per product 3 * 2^-18 = 1.1e-5,
// ~4e-6 typical -- inside the 1e-4 parity budget (tests hold the layer and model outputs to it).
//
// Persistent workgroups (one per CU): the split weight (J*R*4 bytes, <= 128 KB) sits in LDS for the
// whole launch, already in MFMA B-fragment order; each wave streams 32-row tiles of A straight from
// global memory (two 16-byte loads per lane and k-step, next k-step prefetched), splits them in
// registers and issues 3 MFMAs (v_mfma_f32_32x32x16_bf16) per 32-column tile and k-step.
// HBM-bound: reads N*R*4, writes N*J*4.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTgMaxLds = 128 * 1024;      // split weight image

// Bt [J,R] fp32 -> frag[kstep][tile][hi|lo][lane][8] bf16: lane l of tile t, k-step s holds
// Bt[32 t + (l & 31)][16 s + 8 (l >> 5) + j], j = 0..7  (B operand of v_mfma_f32_32x32x16_bf16)
__global__ void tallgemm_split_weight_kernel(const float* __restrict__ bt, bf16x8* __restrict__ frag,
                                             int J, int R) {
  const int tiles = J / 32, ksteps = R / 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (kstep, tile, lane)
  if (idx >= ksteps * tiles * 64) return;
  const int lane = idx & 63, t = (idx >> 6) % tiles, s = (idx >> 6) / tiles;
  const float* src = bt + (size_t)(32 * t + (lane & 31)) * R + 16 * s + 8 * (lane >> 5);
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = src[j];
    const __bf16 h = (__bf16)v;
    hi[j] = h;
    lo[j] = (__bf16)(v - (float)h);
  }
  const size_t base = ((size_t)(s * tiles + t) * 2) * 64 + lane;
  frag[base] = hi;
  frag[base + 64] = lo;
}

struct TgArgs {
  const float* a; const bf16x8* wfrag; const float* bias; float* c;
  int N; int R; int J;
};

template <int JT>       // 32-column tiles per wave = J / 32
__global__ __launch_bounds__(kBlock) void tallgemm_kernel(const TgArgs p) {
  extern __shared__ bf16x8 wlds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int ksteps = p.R / 16;
  const int r31 = lane & 31, h = lane >> 5;

  // the split weight, once per workgroup
  const int n_frag = ksteps * JT * 2 * 64;
  for (int i = threadIdx.x; i < n_frag; i += kBlock) wlds[i] = p.wfrag[i];
  __syncthreads();

  float bias[JT];
#pragma unroll
  for (int t = 0; t < JT; ++t) bias[t] = p.bias ? p.bias[32 * t + r31] : 0.f;

  const int n_tiles = (p.N + 31) / 32;                       // 32-row tiles, dealt round robin to the waves
  for (int tile = blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += gridDim.x * kWavesPerBlock) {
    const int row0 = tile * 32;
    const int arow = min(row0 + r31, p.N - 1);                 // rows past N re-read the last row, never stored
    const float* ap = p.a + (size_t)arow * p.R + 8 * h;

    f32x16 acc[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 n0 = *reinterpret_cast<const float4*>(ap);
    float4 n1 = *reinterpret_cast<const float4*>(ap + 4);
    for (int s = 0; s < ksteps; ++s) {
      const float v[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
      if (s + 1 < ksteps) {                                    // prefetch the next k-step of this row
        n0 = *reinterpret_cast<const float4*>(ap + 16 * (s + 1));
        n1 = *reinterpret_cast<const float4*>(ap + 16 * (s + 1) + 4);
      }
      bf16x8 ahi, alo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 hh = (__bf16)v[j];
        ahi[j] = hh;
        alo[j] = (__bf16)(v[j] - (float)hh);
      }
      const bf16x8* wf = wlds + (size_t)(s * JT) * 2 * 64 + lane;
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        const bf16x8 bhi = wf[t * 128];
        const bf16x8 blo = wf[t * 128 + 64];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[t], 0, 0, 0);
      }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.N) p.c[(size_t)row * p.J + 32 * t + r31] = acc[t][r] + bias[t];
      }
  }
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_tallgemm_supported(int64_t N, int64_t R, int64_t J) {
  const bool j_ok = (J == 32 || J == 64 || J == 128 || J == 256);
  return (N > 0 && R >= 16 && R % 16 == 0 && R <= 1024 && j_ok && R * J * 4 <= kTgMaxLds) ? 1 : 0;
}

// workspace: the split weight image, R*J*4 bytes
extern "C" int64_t mlgnn_tallgemm_workspace_bytes(int64_t R, int64_t J) {
  if (R <= 0 || J <= 0) return MLGNN_E_SHAPE;
  return R * J * 4;
}

extern "C" int mlgnn_tallgemm_nt(const void* a, const void* bt, const float* bias, void* c, void* workspace,
                                 int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, int dtype,
                                 void* stream) {
  if (dtype != MLGNN_DTYPE_F32) return MLGNN_E_DTYPE;
  if (N < 0 || N > INT32_MAX) return MLGNN_E_SHAPE;
  if (N == 0) return 0;
  if (!mlgnn_tallgemm_supported(N, R, J)) return MLGNN_E_SHAPE;
  if (!a || !bt || !c || !workspace) return MLGNN_E_NULL;
  if (workspace_bytes < R * J * 4) return MLGNN_E_WORKSPACE;
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0) return MLGNN_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int n_frag_lanes = (int)(R / 16) * (int)(J / 32) * 64;
  hipLaunchKernelGGL(tallgemm_split_weight_kernel, dim3((n_frag_lanes + 255) / 256), dim3(256), 0, s,
                     (const float*)bt, (bf16x8*)workspace, (int)J, (int)R);
  int err = (int)hipGetLastError();
  if (err) return err;
  TgArgs p;
  p.a = (const float*)a; p.wfrag = (const bf16x8*)workspace; p.bias = bias; p.c = (float*)c;
  p.N = (int)N; p.R = (int)R; p.J = (int)J;
  const size_t lds = (size_t)R * J * 4;
  const int64_t tiles = (N + 31) / 32;
  int grid = (int)((tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  if (grid > 256) grid = 256;                      // persistent: one workgroup per CU
  const dim3 g(grid), b(kBlock);
#define MLGNN_TG_LAUNCH(JT_)                                                                          \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tallgemm_kernel<JT_>),                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, kTgMaxLds);                   \
  hipLaunchKernelGGL((tallgemm_kernel<JT_>), g, b, lds, s, p);
  switch (J / 32) {
    case 1: MLGNN_TG_LAUNCH(1) break;
    case 2: MLGNN_TG_LAUNCH(2) break;
    case 4: MLGNN_TG_LAUNCH(4) break;
    default: MLGNN_TG_LAUNCH(8) break;
  }
#undef MLGNN_TG_LAUNCH
  return (int)hipGetLastError();
}
