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
#include "tile_gemm.h"

namespace mlgnn {

constexpr int kDpMaxN = 160, kDpMaxK = 48, kDpMaxC = 64;
constexpr int kDpSK = kDpMaxK + 1;      // odd strides: row reads and transposed reads both stay
constexpr int kDpSC = kDpMaxC + 1;      // (nearly) bank-conflict free
constexpr float kDpEps = 1e-15f;
// 16 waves per pooled graph (see densesage.hip): independent latency-bound tiles spread over more waves
constexpr int kDpBlock = 1024, kDpWaves = kDpBlock / kWave;

struct DpArgs {                    // z, adj, logits, s_out, x_out, a_out: T (fp32 or bf16 storage); partial: fp32
  const void* z; const void* adj; const void* logits;
  void* s_out; void* x_out; void* a_out; float* partial;   // partial[b] = {sum (A - S S^T)^2, sum entropy}
  int N; int K; int C; int adj_batched;
};

template <typename ST>
__global__ __launch_bounds__(kDpBlock) void diffpool_fwd_kernel(const DpArgs p) {
  __shared__ float S[kDpMaxN][kDpSK];
  __shared__ float T[kDpMaxN][kDpSK];
  __shared__ float Z[kDpMaxN][kDpSC];
  __shared__ float red[kDpWaves][2];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const int N = p.N, K = p.K, C = p.C;
  const StoredIn<ST> zb{static_cast<const ST*>(p.z) + (size_t)b * N * C};
  const StoredIn<ST> lb{static_cast<const ST*>(p.logits) + (size_t)b * N * K};
  const StoredIn<ST> ab{static_cast<const ST*>(p.adj) + (p.adj_batched ? (size_t)b * N * N : 0)};
  const int NP = (N + 15) & ~15, KP = (K + 15) & ~15;

  // ---- softmax rows -> S (zero padded), entropy; Z -> LDS --------------------------------------
  float ent = 0.f;
  for (int r = tid; r < NP; r += kDpBlock) {
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
        stored_write<ST>(p.s_out, ((size_t)b * N + r) * K + k, s);
      }
      for (int k = K; k < KP; ++k) S[r][k] = 0.f;
    } else {
      for (int k = 0; k < KP; ++k) S[r][k] = 0.f;
    }
  }
  for (int idx = tid; idx < N * C; idx += kDpBlock) Z[idx / C][idx % C] = zb[idx];
  __syncthreads();

  const int l15 = lane & 15, lq = lane >> 4;
  const int Nt = NP / 16, Kt = KP / 16, Ct = (C + 15) / 16;

  // ---- X' = S^T Z  [K, C] ----------------------------------------------------------------------
  for (int t = wave; t < Kt * Ct; t += kDpWaves) {
    const int i0 = (t / Ct) * 16, j0 = (t % Ct) * 16;
    const f32x4 acc = tile_gemm(N,
        [&](int i, int k) { return k < N ? S[k][i0 + i] : 0.f; },
        [&](int k, int j) { return (k < N && j0 + j < C) ? Z[k][j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = j0 + l15;
      if (row < K && col < C) stored_write<ST>(p.x_out, ((size_t)b * K + row) * C + col, acc[r]);
    }
  }

  // ---- T = A S  [N, K] (kept in LDS) ------------------------------------------------------------
  for (int t = wave; t < Nt * Kt; t += kDpWaves) {
    const int i0 = (t / Kt) * 16, j0 = (t % Kt) * 16;
    const f32x4 acc = tile_gemm(N,
        [&](int i, int k) { return (i0 + i < N && k < N) ? ab[(size_t)(i0 + i) * N + k] : 0.f; },
        [&](int k, int j) { return k < N ? S[k][j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) T[i0 + lq * 4 + r][j0 + l15] = acc[r];
  }
  __syncthreads();

  // ---- A' = S^T T  [K, K] -----------------------------------------------------------------------
  for (int t = wave; t < Kt * Kt; t += kDpWaves) {
    const int i0 = (t / Kt) * 16, j0 = (t % Kt) * 16;
    const f32x4 acc = tile_gemm(N,
        [&](int i, int k) { return k < N ? S[k][i0 + i] : 0.f; },
        [&](int k, int j) { return k < N ? T[k][j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = j0 + l15;
      if (row < K && col < K) stored_write<ST>(p.a_out, ((size_t)b * K + row) * K + col, acc[r]);
    }
  }

  // ---- sum (A - S S^T)^2 ------------------------------------------------------------------------
  float sq = 0.f;
  for (int t = wave; t < Nt * Nt; t += kDpWaves) {
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
    for (int w = 0; w < kDpWaves; ++w) { a0 += red[w][0]; a1 += red[w][1]; }
    p.partial[2 * b] = a0;
    p.partial[2 * b + 1] = a1;
  }
}

// ------------------------------------------------------------------------------------------------
// backward of the contraction, same one-workgroup-per-pooled-graph layout.  With S the saved softmax,
// g = grad of X', h = grad of A', cl = grad_link / (numel * ||A - S S^T||_F), ce = grad_ent / (B N):
//   dZ = S g
//   dS = Z g^T + (A S) h^T + (A^T S) h - cl (A S + A^T S - 2 S (S^T S)) - ce (log(S + eps) + S / (S + eps))
//   ds = S * (dS - rowsum(dS * S))                                   (softmax backward)
//   dA = (S h) S^T + cl (A - S S^T)                                  (only when the adjacency needs it)
// ((D + D^T) S with D = A - S S^T is expanded so that the [N,N] matrix D never has to be stored.)
// LDS: S, A S, A^T S, dS (4 x 31 KB) + S^T S (9 KB); Z, g, h and A are read from global memory
// directly in MFMA operand layout.
// ------------------------------------------------------------------------------------------------
struct DpBwdArgs {                 // everything T except coef (fp32)
  const void* z; const void* adj; const void* s; const void* gx; const void* ga; const float* coef;
  void* gz; void* gs; void* gadj;
  int N; int K; int C; int adj_batched;
};

template <typename ST>
__global__ __launch_bounds__(kDpBlock) void diffpool_bwd_kernel(const DpBwdArgs p) {
  __shared__ float S[kDpMaxN][kDpSK];
  __shared__ float AS[kDpMaxN][kDpSK];
  __shared__ float AtS[kDpMaxN][kDpSK];
  __shared__ float GS[kDpMaxN][kDpSK];
  __shared__ float SS[kDpMaxK][kDpSK];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const int N = p.N, K = p.K, C = p.C;
  const StoredIn<ST> zb{static_cast<const ST*>(p.z) + (size_t)b * N * C};
  const StoredIn<ST> sb{static_cast<const ST*>(p.s) + (size_t)b * N * K};
  const StoredIn<ST> ab{static_cast<const ST*>(p.adj) + (p.adj_batched ? (size_t)b * N * N : 0)};
  const StoredIn<ST> gxb{static_cast<const ST*>(p.gx) + (size_t)b * K * C};
  const StoredIn<ST> gab{static_cast<const ST*>(p.ga) + (size_t)b * K * K};
  const float cl = p.coef[0], ce = p.coef[1];
  const int NP = (N + 15) & ~15, KP = (K + 15) & ~15;
  const int Nt = NP / 16, Kt = KP / 16, Ct = (C + 15) / 16;
  const int l15 = lane & 15, lq = lane >> 4;

  for (int idx = tid; idx < NP * KP; idx += kDpBlock) {
    const int r = idx / KP, k = idx % KP;
    S[r][k] = (r < N && k < K) ? sb[(size_t)r * K + k] : 0.f;
  }
  __syncthreads();

  // ---- A S, A^T S  [N,K]  and  S^T S  [K,K] ------------------------------------------------------
  for (int t = wave; t < 2 * Nt * Kt + Kt * Kt; t += kDpWaves) {
    if (t < 2 * Nt * Kt) {
      const bool tr = t >= Nt * Kt;
      const int tt = tr ? t - Nt * Kt : t;
      const int i0 = (tt / Kt) * 16, j0 = (tt % Kt) * 16;
      const f32x4 acc = tile_gemm(N,
          [&](int i, int k) {
            if (i0 + i >= N || k >= N) return 0.f;
            return tr ? ab[(size_t)k * N + i0 + i] : ab[(size_t)(i0 + i) * N + k];
          },
          [&](int k, int j) { return k < N ? S[k][j0 + j] : 0.f; });
#pragma unroll
      for (int r = 0; r < 4; ++r) (tr ? AtS : AS)[i0 + lq * 4 + r][j0 + l15] = acc[r];
    } else {
      const int tt = t - 2 * Nt * Kt;
      const int i0 = (tt / Kt) * 16, j0 = (tt % Kt) * 16;
      const f32x4 acc = tile_gemm(N,
          [&](int i, int k) { return k < N ? S[k][i0 + i] : 0.f; },
          [&](int k, int j) { return k < N ? S[k][j0 + j] : 0.f; });
#pragma unroll
      for (int r = 0; r < 4; ++r) SS[i0 + lq * 4 + r][j0 + l15] = acc[r];
    }
  }
  __syncthreads();

  // ---- dS tiles [N,K] -----------------------------------------------------------------------------
  for (int t = wave; t < Nt * Kt; t += kDpWaves) {
    const int i0 = (t / Kt) * 16, j0 = (t % Kt) * 16;
    f32x4 acc = tile_gemm(C,                                                     // Z g^T
        [&](int i, int k) { return (i0 + i < N && k < C) ? zb[(size_t)(i0 + i) * C + k] : 0.f; },
        [&](int k, int j) { return (j0 + j < K && k < C) ? gxb[(size_t)(j0 + j) * C + k] : 0.f; });
    const f32x4 t1 = tile_gemm(K,                                                // (A S) h^T
        [&](int i, int k) { return k < KP ? AS[i0 + i][k] : 0.f; },
        [&](int k, int j) { return (j0 + j < K && k < K) ? gab[(size_t)(j0 + j) * K + k] : 0.f; });
    const f32x4 t2 = tile_gemm(K,                                                // (A^T S) h
        [&](int i, int k) { return k < KP ? AtS[i0 + i][k] : 0.f; },
        [&](int k, int j) { return (j0 + j < K && k < K) ? gab[(size_t)k * K + j0 + j] : 0.f; });
    const f32x4 t3 = tile_gemm(K,                                                // S (S^T S)
        [&](int i, int k) { return k < KP ? S[i0 + i][k] : 0.f; },
        [&](int k, int j) { return k < KP ? SS[k][j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = j0 + l15;
      const float sv = S[row][col];
      float g = acc[r] + t1[r] + t2[r] - cl * (AS[row][col] + AtS[row][col] - 2.f * t3[r]);
      g -= ce * (__logf(sv + kDpEps) + sv / (sv + kDpEps));
      GS[row][col] = (row < N && col < K) ? g : 0.f;
    }
  }
  __syncthreads();

  // ---- softmax backward, one thread per node row -------------------------------------------------
  for (int r = tid; r < N; r += kDpBlock) {
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot = fmaf(GS[r][k], S[r][k], dot);
    for (int k = 0; k < K; ++k) stored_write<ST>(p.gs, ((size_t)b * N + r) * K + k, S[r][k] * (GS[r][k] - dot));
  }

  // ---- dZ = S g  [N,C] ----------------------------------------------------------------------------
  for (int t = wave; t < Nt * Ct; t += kDpWaves) {
    const int i0 = (t / Ct) * 16, j0 = (t % Ct) * 16;
    const f32x4 acc = tile_gemm(K,
        [&](int i, int k) { return k < KP ? S[i0 + i][k] : 0.f; },
        [&](int k, int j) { return (k < K && j0 + j < C) ? gxb[(size_t)k * C + j0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = j0 + l15;
      if (row < N && col < C) stored_write<ST>(p.gz, ((size_t)b * N + row) * C + col, acc[r]);
    }
  }

  // ---- dA = (S h) S^T + cl (A - S S^T)  [N,N] -------------------------------------------------------
  if (p.gadj) {
    __syncthreads();                       // AS is free now: reuse it for P = S h
    for (int t = wave; t < Nt * Kt; t += kDpWaves) {
      const int i0 = (t / Kt) * 16, j0 = (t % Kt) * 16;
      const f32x4 acc = tile_gemm(K,
          [&](int i, int k) { return k < KP ? S[i0 + i][k] : 0.f; },
          [&](int k, int j) { return (k < K && j0 + j < K) ? gab[(size_t)k * K + j0 + j] : 0.f; });
#pragma unroll
      for (int r = 0; r < 4; ++r) AS[i0 + lq * 4 + r][j0 + l15] = acc[r];
    }
    __syncthreads();
    const size_t gb_off = (size_t)b * N * N;
    for (int t = wave; t < Nt * Nt; t += kDpWaves) {
      const int i0 = (t / Nt) * 16, j0 = (t % Nt) * 16;
      const f32x4 pst = tile_gemm(K,
          [&](int i, int k) { return k < KP ? AS[i0 + i][k] : 0.f; },
          [&](int k, int j) { return k < KP ? S[j0 + j][k] : 0.f; });
      const f32x4 sst = tile_gemm(K,
          [&](int i, int k) { return k < KP ? S[i0 + i][k] : 0.f; },
          [&](int k, int j) { return k < KP ? S[j0 + j][k] : 0.f; });
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + lq * 4 + r, col = j0 + l15;
        if (row < N && col < N)
          stored_write<ST>(p.gadj, gb_off + (size_t)row * N + col, pst[r] + cl * (ab[(size_t)row * N + col] - sst[r]));
      }
    }
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
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (B < 0 || B > INT32_MAX || !mlgnn_diffpool_fwd_supported(N, K, C)) return MLGNN_E_SHAPE;
  if (B == 0) return 0;
  if (!z || !adj || !s_logits || !s_out || !x_out || !adj_out || !partial) return MLGNN_E_NULL;
  DpArgs a;
  a.z = z; a.adj = adj; a.logits = s_logits;
  a.s_out = s_out; a.x_out = x_out; a.a_out = adj_out; a.partial = partial;
  a.N = (int)N; a.K = (int)K; a.C = (int)C; a.adj_batched = adj_batched;
  if (dtype == MLGNN_DTYPE_BF16) hipLaunchKernelGGL(diffpool_fwd_kernel<bf16_t>, dim3((unsigned)B), dim3(kDpBlock), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(diffpool_fwd_kernel<float>, dim3((unsigned)B), dim3(kDpBlock), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_diffpool_bwd(const void* z, const void* adj, const void* s_softmax, const void* grad_x,
                                  const void* grad_adj_out, const float* coef, void* grad_z, void* grad_s,
                                  void* grad_adj, int64_t B, int64_t N, int64_t K, int64_t C,
                                  int adj_batched, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (B < 0 || B > INT32_MAX || !mlgnn_diffpool_fwd_supported(N, K, C)) return MLGNN_E_SHAPE;
  if (B == 0) return 0;
  if (!z || !adj || !s_softmax || !grad_x || !grad_adj_out || !coef || !grad_z || !grad_s) return MLGNN_E_NULL;
  DpBwdArgs a;
  a.z = z; a.adj = adj; a.s = s_softmax;
  a.gx = grad_x; a.ga = grad_adj_out; a.coef = coef;
  a.gz = grad_z; a.gs = grad_s; a.gadj = grad_adj;
  a.N = (int)N; a.K = (int)K; a.C = (int)C; a.adj_batched = adj_batched;
  if (dtype == MLGNN_DTYPE_BF16) hipLaunchKernelGGL(diffpool_bwd_kernel<bf16_t>, dim3((unsigned)B), dim3(kDpBlock), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(diffpool_bwd_kernel<float>, dim3((unsigned)B), dim3(kDpBlock), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
