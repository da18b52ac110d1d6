// Linear layer over a handful of rows with a very long input: y [M, J] = x [M, K] W[J, K]^T + b with M <= 64 samples and
// K ~ 1e5 -- the first layer of MultilevelGNN's head (reference: models/multilevel_gnn.py:121-127, config/kirc.yaml:
// nn.Linear(64 * 146 * 9 = 84 096, 512) over a batch of 64 samples).  The weight (172 MB) is the only large operand of all
// three products of the layer and is used once per sample: the layer is a STREAM over W (forward, input gradient) or over
// dW (weight gradient), 22 us each at 8 TB/s, and nothing like the square GEMMs a library tunes for (measured: 189 + 108 +
// 89 us).  Three kernels, plain fp32 arithmetic on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains; 2 M J K =
// 5.5 GFLOP per product = 35 us of the 157 TFLOP/s pipe, next to the 22 us stream), tiles staged through LDS, every global
// access a 16-byte one along K.  (The first version used packed VALU FMAs on 4x4 register tiles and was bound by LDS
// bandwidth -- 8 x 16-byte LDS reads per 32 packed FMAs: 166 / 103 / 103 us; an MFMA takes its operands from LDS once per wave.)
//
//   forward   workgroup = (64 columns of J) x (a range of K); partial [M, 64] sums per K range, reduced in a fixed order
//   dX        workgroup = 128 columns of K, the whole J range inside: dX[M, 128] written once
//   dW        workgroup = (64 rows of J) x (128 columns of K), the whole M range inside: dW tile written once
//
// Deterministic: no atomics, fixed summation orders.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

constexpr int kSkM = 64;            // rows (samples) a tile holds; fewer are zero padded
constexpr int kSkJ = 64;            // J tile
constexpr int kSkK = 128;           // K tile
constexpr int kSkLd = kSkK + 4;     // LDS row stride (floats): 16-byte aligned rows, neighbouring rows 4 banks apart
constexpr int kSkThreads = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// C/D: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
__device__ __forceinline__ int sk_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// global [rows, ld] tile (rows x 128 floats from column k0, zero past `rows_valid` rows / `k_valid` columns) -> LDS [64][132]
__device__ __forceinline__ void sk_load_tile(float* __restrict__ lds, const float* __restrict__ src, int64_t ld, int row0,
                                             int rows_valid, int k0, int k_valid) {
  // 64 rows x 32 float4: 8 per thread, consecutive threads along K (512-byte row segments)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int u = threadIdx.x + i * kSkThreads;
    const int r = u >> 5, q = u & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows_valid && k0 + 4 * q < k_valid) v = *reinterpret_cast<const float4*>(src + (int64_t)(row0 + r) * ld + k0 + 4 * q);
    *reinterpret_cast<float4*>(lds + r * kSkLd + 4 * q) = v;
  }
}

// ---- forward: partial[split][m][j] = sum_{k in range(split)} x[m, k] W[j, k] -----------------------------------------------
__global__ __launch_bounds__(kSkThreads) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ partial,
                                                               int M, int J, int K, int k_per_split) {
  extern __shared__ __attribute__((aligned(16))) float sk_smem[];          // 2 x 33 KB: past the 64 KB of static LDS
  float* xs = sk_smem;
  float* ws = sk_smem + kSkM * kSkLd;
  const int jt = blockIdx.x, split = blockIdx.y;
  const int k_begin = split * k_per_split, k_end = min(K, k_begin + k_per_split);
  // 4 waves: wave (wm, wj) owns the 32 x 32 block (rows 32 wm.., columns 32 wj..) of the [64 m, 64 j] tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wj = wave & 1, i31 = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = k_begin; k0 < k_end; k0 += kSkK) {
    __syncthreads();
    sk_load_tile(xs, x, K, 0, M, k0, k_end);
    sk_load_tile(ws, w, K, jt * kSkJ, min(kSkJ, J - jt * kSkJ), k0, k_end);
    __syncthreads();
    const float* ap = xs + (32 * wm + i31) * kSkLd + 4 * h;
    const float* bp = ws + (32 * wj + i31) * kSkLd + 4 * h;
    // 8 columns of K per 16-byte read pair: step s contracts k = k8 + s (half 0) with k8 + 4 + s (half 1) -- any pairing
    // works as long as both operands use the same one
#pragma unroll 4
    for (int k8 = 0; k8 < kSkK; k8 += 8) {
      const float4 a = *reinterpret_cast<const float4*>(ap + k8);
      const float4 b = *reinterpret_cast<const float4*>(bp + k8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  }
  float* out = partial + (size_t)split * M * J;
  const int j = jt * kSkJ + 32 * wj + i31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = 32 * wm + sk_crow(r, h);
    if (m < M && j < J) out[(size_t)m * J + j] = acc[r] + ((split == 0 && bias) ? bias[j] : 0.f);   // (the bias enters once)
  }
}

// ---- input gradient: dx[m, k] = sum_j dy[m, j] W[j, k]; workgroup = 128 columns of K ------------------------------------------
__global__ __launch_bounds__(kSkThreads) void skinny_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int M, int J, int K) {
  __shared__ __attribute__((aligned(16))) float ws[kSkJ * kSkLd];          // W[j tile][k tile]
  __shared__ __attribute__((aligned(16))) float ds[kSkM * (kSkJ + 4)];     // dy[m][j tile]
  const int k0 = blockIdx.x * kSkK;
  // 4 waves: wave (wm, wk) owns rows 32 wm.. and columns 64 wk.. of the [64 m, 128 k] tile = two 32 x 32 blocks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wk = wave & 1, i31 = lane & 31, h = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int j0 = 0; j0 < J; j0 += kSkJ) {
    __syncthreads();
    sk_load_tile(ws, w, K, j0, min(kSkJ, J - j0), k0, K);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int u = threadIdx.x + i * kSkThreads;
      const int m = u >> 6, j = u & 63;
      ds[m * (kSkJ + 4) + j] = (m < M && j0 + j < J) ? dy[(size_t)m * J + j0 + j] : 0.f;
    }
    __syncthreads();
    const float* ap = ds + (32 * wm + i31) * (kSkJ + 4) + 4 * h;          // A[m][j]: 4 consecutive j per read
    const float* bp = ws + (4 * h) * kSkLd + 64 * wk + i31;                // B[j][k column]
#pragma unroll 2
    for (int j8 = 0; j8 < kSkJ; j8 += 8) {
      const float4 a = *reinterpret_cast<const float4*>(ap + j8);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {                                     // contracts j = j8 + s2 (half 0), j8 + 4 + s2 (half 1)
        const float b0 = bp[(j8 + s2) * kSkLd], b1 = bp[(j8 + s2) * kSkLd + 32];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], b1, acc[1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = k0 + 64 * wk + 32 * t + i31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * wm + sk_crow(r, h);
      if (m < M && k < K) dx[(size_t)m * K + k] = acc[t][r];
    }
  }
}

// ---- weight gradient: dw[j, k] = sum_m dy[m, j] x[m, k]; workgroup = 64 rows of J x 128 columns of K ---------------------------
__global__ __launch_bounds__(kSkThreads) void skinny_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ dw, float* __restrict__ db, int M, int J,
                                                              int K) {
  __shared__ __attribute__((aligned(16))) float xs[kSkM * kSkLd];          // x[m][k tile]
  __shared__ __attribute__((aligned(16))) float ds[kSkM * (kSkJ + 4)];     // dy[m][j tile]
  const int k0 = blockIdx.x * kSkK, j0 = blockIdx.y * kSkJ;
  sk_load_tile(xs, x, K, 0, M, k0, K);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int u = threadIdx.x + i * kSkThreads;
    const int m = u >> 6, j = u & 63;
    ds[m * (kSkJ + 4) + j] = (m < M && j0 + j < J) ? dy[(size_t)m * J + j0 + j] : 0.f;
  }
  __syncthreads();
  // (packed VALU FMAs on 4 x 8 register tiles with 16-byte stores: 103 us; the fp32-MFMA form of this product -- 64 MFMAs per
  // wave between a 48 KB tile load and a 32 KB store of 4-byte accumulator registers -- measured 126 us)
  // the bias gradient rides along: the workgroups of the first K tile hold every dy tile once
  if (db && blockIdx.x == 0 && threadIdx.x < kSkJ && j0 + (int)threadIdx.x < J) {
    float t = 0.f;
    for (int m = 0; m < kSkM; ++m) t += ds[m * (kSkJ + 4) + threadIdx.x];
    db[j0 + threadIdx.x] = t;
  }
  const int ji = threadIdx.x >> 4, ki = threadIdx.x & 15;            // 4 rows of J x 8 columns of K per thread
  float acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[r][c] = 0.f;
#pragma unroll 4
  for (int m = 0; m < kSkM; ++m) {
    const float4 d = *reinterpret_cast<const float4*>(ds + m * (kSkJ + 4) + 4 * ji);
    const float4 x0 = *reinterpret_cast<const float4*>(xs + m * kSkLd + 4 * ki);
    const float4 x1 = *reinterpret_cast<const float4*>(xs + m * kSkLd + 64 + 4 * ki);
    const float dv[4] = {d.x, d.y, d.z, d.w};
    const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[r][c] = fmaf(dv[r], xv[c], acc[r][c]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = j0 + 4 * ji + r;
    if (j >= J) continue;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int k = k0 + 64 * half + 4 * ki;
      if (k < K) {
        using f4 = __attribute__((ext_vector_type(4))) float;
        const f4 t = {acc[r][4 * half], acc[r][4 * half + 1], acc[r][4 * half + 2], acc[r][4 * half + 3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(dw + (size_t)j * K + k));     // written once, read by the optimizer later
      }
    }
  }
}

// column sums of dy [M, J] -> db [J]
__global__ __launch_bounds__(256) void skinny_db_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int J) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= J) return;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += dy[(size_t)m * J + j];
  db[j] = s;
}

static int sk_splits(int64_t K) {
  // K ranges of whole 128-column tiles; about 2 workgroups per CU over the (J tiles x splits) grid is decided by the caller
  return (int)((K + kSkK - 1) / kSkK);
}

}  // namespace mlgnn

using namespace mlgnn;

static bool sk_ok(int64_t M, int64_t J, int64_t K) {
  return M >= 1 && M <= kSkM && J >= 1 && J <= 65535 * (int64_t)kSkJ && K >= 4 && K % 4 == 0 && K <= INT32_MAX &&
         M * K <= INT32_MAX && J * K / 4 <= INT32_MAX;
}

extern "C" int mlgnn_skinny_linear_supported(int64_t M, int64_t J, int64_t K) { return sk_ok(M, J, K) ? 1 : 0; }

// K ranges per J tile: enough workgroups to fill the chip twice, never below 8 tiles of K per workgroup
static int sk_plan(int64_t J, int64_t K, int* k_per_split) {
  const int tiles = sk_splits(K);
  const int jt = (int)((J + kSkJ - 1) / kSkJ);
  int splits = (512 + jt - 1) / jt;
  if (splits > tiles / 8) splits = tiles / 8;
  if (splits < 1) splits = 1;
  const int tiles_per = (tiles + splits - 1) / splits;
  *k_per_split = tiles_per * kSkK;
  return (tiles + tiles_per - 1) / tiles_per;
}

extern "C" int64_t mlgnn_skinny_linear_fwd_workspace_floats(int64_t M, int64_t J, int64_t K) {
  if (!sk_ok(M, J, K)) return MLGNN_E_SHAPE;
  int kps;
  return (int64_t)sk_plan(J, K, &kps) * M * J;
}

extern "C" int mlgnn_skinny_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* workspace,
                                       int64_t workspace_floats, int64_t M, int64_t J, int64_t K, void* stream) {
  if (!sk_ok(M, J, K)) return MLGNN_E_SHAPE;
  if (!x || !w || !y || !workspace) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) != 0) return MLGNN_E_ALIGN;
  int kps;
  const int splits = sk_plan(J, K, &kps);
  if (workspace_floats < (int64_t)splits * M * J) return MLGNN_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  constexpr int lds_bytes = (kSkM + kSkJ) * kSkLd * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(skinny_fwd_kernel, dim3((unsigned)((J + kSkJ - 1) / kSkJ), (unsigned)splits), dim3(kSkThreads), lds_bytes, s,
                     x, w, bias, workspace, (int)M, (int)J, (int)K, kps);
  launch_reduce_partials(workspace, y, splits, (int)(M * J), s);          // K ranges in order: bitwise reproducible
  return (int)hipGetLastError();
}

extern "C" int mlgnn_skinny_linear_bwd(const float* grad_out, const float* x, const float* w, float* grad_x, float* grad_w,
                                       float* grad_b, int64_t M, int64_t J, int64_t K, void* stream) {
  if (!sk_ok(M, J, K)) return MLGNN_E_SHAPE;
  if (!grad_out || (grad_x && !w) || (grad_w && !x)) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(grad_x) |
        reinterpret_cast<uintptr_t>(grad_w)) & 15) != 0)
    return MLGNN_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const unsigned kt = (unsigned)((K + kSkK - 1) / kSkK);
  if (grad_x) hipLaunchKernelGGL(skinny_dx_kernel, dim3(kt), dim3(kSkThreads), 0, s, grad_out, w, grad_x, (int)M, (int)J, (int)K);
  if (grad_w)
    hipLaunchKernelGGL(skinny_dw_kernel, dim3(kt, (unsigned)((J + kSkJ - 1) / kSkJ)), dim3(kSkThreads), 0, s, grad_out, x, grad_w,
                       grad_b, (int)M, (int)J, (int)K);
  else if (grad_b)
    hipLaunchKernelGGL(skinny_db_kernel, dim3((unsigned)((J + 255) / 256)), dim3(256), 0, s, grad_out, grad_b, (int)M, (int)J);
  return (int)hipGetLastError();
}
