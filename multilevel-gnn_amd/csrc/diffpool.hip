// DiffPool soft-assignment contraction, fused forward on the fp32 matrix cores.
//
// Reference: torch_geometric.nn.dense_diff_pool as called from models/diff_pooling.py:64
// (DiffPoolLayer.forward):   S = softmax(s, -1);  X' = S^T Z;  A' = S^T A S;
//   link = ||A - S S^T||_F / numel(A);  ent = mean_n( sum_k -S log(S + 1e-15) )
// -- softmax + 4 batched GEMMs + 2 reductions (~12 launches in the reference, each latency bound at
// 146 x 37).  Here ONE workgroup per batch element keeps S, Z and T = A S in LDS and runs every
// product on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, so 1e-4 parity holds): operands are read
// straight in MFMA layout (lane l: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; C/D: col = l&15,
// row = 4*(l>>4) + reg).  The adjacency is read from global memory (shared by the whole batch at
// level 0, hence L2 resident).
//
// Limits of the fused kernel: N <= 160, K <= 48, C <= 64 (LDS: S 31 KB + T 31 KB + Z 41 KB).
// Larger pooled graphs (BASELINE configs[4]) take library GEMMs.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kDpMaxN = 160, kDpMaxK = 48, kDpMaxC = 64;
constexpr int kDpSK = kDpMaxK + 1;      // odd strides: row reads and transposed reads both stay
constexpr int kDpSC = kDpMaxC + 1;      // (nearly) bank-conflict free
constexpr float kDpEps = 1e-15f;

struct DpArgs {
  const float* z; const float* adj; const float* logits;
  float* s_out; float* x_out; float* a_out; float* partial;   // partial[b] = {sum (A - S S^T)^2, sum entropy}
  int N; int K; int C; int adj_batched;
};

// one 16x16 output tile: acc[i][j] = sum_k a_at(i, k) * b_at(k, j)
template <typename FA, typename FB>
__device__ __forceinline__ f32x4 tile_gemm(int kdim, FA a_at, FB b_at) {
  const int lane = threadIdx.x & (kWave - 1);
  const int l15 = lane & 15, lk = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kdim; k0 += 4) {
    const float a = a_at(l15, k0 + lk);
    const float b = b_at(k0 + lk, l15);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(kBlock) void diffpool_fwd_kernel(const DpArgs p) {
  __shared__ float S[kDpMaxN][kDpSK];
  __shared__ float T[kDpMaxN][kDpSK];
  __shared__ float Z[kDpMaxN][kDpSC];
  __shared__ float red[kWavesPerBlock][2];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const int N = p.N, K = p.K, C = p.C;
  const float* zb = p.z + (size_t)b * N * C;
  const float* lb = p.logits + (size_t)b * N * K;
  const float* ab = p.adj + (p.adj_batched ? (size_t)b * N * N : 0);
  const int NP = (N + 15) & ~15, KP = (K + 15) & ~15;

  // ---- softmax rows -> S (zero padded), entropy; Z -> LDS --------------------------------------
  float ent = 0.f;
  for (int r = tid; r < NP; r += kBlock) {
    if (r < N) {
      float mx = -3.0e38f;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, lb[(size_t)r * K + k]);
      float sum = 0.f;
      for (int k = 0; k < K; ++k) { const float e = __expf(lb[(size_t)r * K + k] - mx); S[r][k] = e; sum += e; }
      const float inv = 1.0f / sum;
      for (int k = 0; k < K; ++k) {
        const float s = S[r][k] * inv;
        S[r][k] = s;
        ent -= s * __logf(s + kDpEps);
        p.s_out[((size_t)b * N + r) * K + k] = s;
      }
      for (int k = K; k < KP; ++k) S[r][k] = 0.f;
    } else {
      for (int k = 0; k < KP; ++k) S[r][k] = 0.f;
    }
  }
  for (int idx = tid; idx < N * C; idx += kBlock) Z[idx / C][idx % C] = zb[idx];
  __syncthreads();

  const int l15 = lane & 15, lq = lane >> 4;
  const int Nt = NP / 16, Kt = KP / 16, Ct = (C + 15) / 16;

  // ---- X' = S^T Z  [K, C] ----------------------------------------------------------------------
  for (int t = wave; t < Kt * Ct; t += kWavesPerBlock) {
    const int i0 = (t / Ct) * 16, j0 = (t % Ct) * 16;
    const f32x4 acc = tile_gemm(N,
        [&](int i, int k) { return k < N ? S[k][i0 + i] : 0.f; },
        [&](int k, int j) { return (k < N && j0 + j < C) ? Z[k][j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = j0 + l15;
      if (row < K && col < C) p.x_out[((size_t)b * K + row) * C + col] = acc[r];
    }
  }

  // ---- T = A S  [N, K] (kept in LDS) ------------------------------------------------------------
  for (int t = wave; t < Nt * Kt; t += kWavesPerBlock) {
    const int i0 = (t / Kt) * 16, j0 = (t % Kt) * 16;
    const f32x4 acc = tile_gemm(N,
        [&](int i, int k) { return (i0 + i < N && k < N) ? ab[(size_t)(i0 + i) * N + k] : 0.f; },
        [&](int k, int j) { return k < N ? S[k][j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) T[i0 + lq * 4 + r][j0 + l15] = acc[r];
  }
  __syncthreads();

  // ---- A' = S^T T  [K, K] -----------------------------------------------------------------------
  for (int t = wave; t < Kt * Kt; t += kWavesPerBlock) {
    const int i0 = (t / Kt) * 16, j0 = (t % Kt) * 16;
    const f32x4 acc = tile_gemm(N,
        [&](int i, int k) { return k < N ? S[k][i0 + i] : 0.f; },
        [&](int k, int j) { return k < N ? T[k][j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = j0 + l15;
      if (row < K && col < K) p.a_out[((size_t)b * K + row) * K + col] = acc[r];
    }
  }

  // ---- sum (A - S S^T)^2 ------------------------------------------------------------------------
  float sq = 0.f;
  for (int t = wave; t < Nt * Nt; t += kWavesPerBlock) {
    const int i0 = (t / Nt) * 16, j0 = (t % Nt) * 16;
    const f32x4 acc = tile_gemm(K,
        [&](int i, int k) { return k < KP ? S[i0 + i][k] : 0.f; },
        [&](int k, int j) { return k < KP ? S[j0 + j][k] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = j0 + l15;
      if (row < N && col < N) { const float dlt = ab[(size_t)row * N + col] - acc[r]; sq = fmaf(dlt, dlt, sq); }
    }
  }

  // ---- block reduction of the two scalars -------------------------------------------------------
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sq += __shfl_xor(sq, off); ent += __shfl_xor(ent, off); }
  if (lane == 0) { red[wave][0] = sq; red[wave][1] = ent; }
  __syncthreads();
  if (tid == 0) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) { a0 += red[w][0]; a1 += red[w][1]; }
    p.partial[2 * b] = a0;
    p.partial[2 * b + 1] = a1;
  }
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_diffpool_fwd_supported(int64_t N, int64_t K, int64_t C) {
  return (N > 0 && K > 0 && C > 0 && N <= kDpMaxN && K <= kDpMaxK && C <= kDpMaxC) ? 1 : 0;
}

extern "C" int mlgnn_diffpool_fwd(const void* z, const void* adj, const void* s_logits, void* s_out,
                                  void* x_out, void* adj_out, float* partial, int64_t B, int64_t N,
                                  int64_t K, int64_t C, int adj_batched, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32) return MLGNN_E_DTYPE;
  if (B < 0 || B > INT32_MAX || !mlgnn_diffpool_fwd_supported(N, K, C)) return MLGNN_E_SHAPE;
  if (B == 0) return 0;
  if (!z || !adj || !s_logits || !s_out || !x_out || !adj_out || !partial) return MLGNN_E_NULL;
  DpArgs a;
  a.z = (const float*)z; a.adj = (const float*)adj; a.logits = (const float*)s_logits;
  a.s_out = (float*)s_out; a.x_out = (float*)x_out; a.a_out = (float*)adj_out; a.partial = partial;
  a.N = (int)N; a.K = (int)K; a.C = (int)C; a.adj_batched = adj_batched;
  hipLaunchKernelGGL(diffpool_fwd_kernel, dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
