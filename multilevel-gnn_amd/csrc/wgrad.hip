// Weight / bias gradient of a Linear layer over a very tall activation matrix:
//     dW[M,K] = A^T B,   db[M] = column sums of A,   A = grad_out [N,M],  B = x [N,K],  N >> M,K
//
// Reference: autograd of nn.Linear inside MLP (models/gcn_lib/sparse/torch_nn.py:54-75, the dense
// epilogue of every conv) -- a "TN" GEMM whose reduction dimension is the 640 000 node rows and
// whose output is 128x256: the library picks a 32x32-tile kernel without split-K (~39 TFLOP/s
// measured).  Here the rows are split over the chip (one slab per workgroup), each wave keeps its
// share of the [M,K] output in fp32 MFMA accumulators (v_mfma_f32_32x32x2_f32: exact fp32 FMA
// chain), operand tiles are double-buffered through LDS, and per-slab partials are summed in a
// fixed order by reduce_partials (bitwise reproducible).
//
// MFMA-bound: 2*N*M*K FLOP at the 157 TFLOP/s fp32-matrix peak; reads A and B once from HBM.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kTile = 32;

struct WgradArgs {
  const float* a; const float* b; float* ws;
  int N; int M; int K; int rows_per_block; int out_cols;   // out_cols = M*K + M
};

// wave layout WM x WK, tiles per wave TM x TK.  Row slab of the workgroup is walked in stages of
// kStageRows rows: both operand tiles of a stage ([rows, M] of grad_out and [rows, K] of x, zero
// padded to the tile grid) are fetched with 16-byte coalesced loads into registers while the previous
// stage is being multiplied out of LDS, then written to the other LDS buffer (classic double
// buffering; one barrier per stage).  MFMA operands are single ds_read_b32 per lane: lane l reads
// tile[2*kk + (l>>5)][col0 + (l&31)] -- 32 consecutive floats per half-wave, conflict free.
constexpr int kStageRows = 32;

template <int WM, int WK, int TM, int TK, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void linear_wgrad_kernel(const WgradArgs p) {
  constexpr int MP = WM * TM * kTile, KP = WK * TK * kTile, W = MP + KP;   // padded operand widths
  constexpr int kChunks = kStageRows * W / 4;                               // float4 chunks per stage
  constexpr int kPerThread = (kChunks + kBlock - 1) / kBlock;
  __shared__ float4 tile4[2][kStageRows * W / 4];

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int wm = wave / WK, wk = wave % WK;
  const int half = lane >> 5, l31 = lane & 31;
  const int m_base = wm * TM * kTile, k_base = wk * TK * kTile;

  f32x16 acc[TM][TK];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) bsum[i] = 0.f;

  const int r_begin = blockIdx.x * p.rows_per_block;
  const int r_end = min(p.N, r_begin + p.rows_per_block);

  float4 stage[kPerThread];
  // global -> registers: chunk c covers columns [4*(c % (W/4)), +4) of stage row c / (W/4).
  // ALIGNED (M % 4 == 0 and K % 4 == 0): straight-line code -- clamped addresses, unconditional 16-byte
  // loads, zeroing by select -- so the loads stay in flight across the multiply (no control flow for the
  // compiler's waitcnt insertion to be conservative about).
  static_assert(kChunks % kBlock == 0, "stage must split evenly over the workgroup");
  auto fetch = [&](int r0) {
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const int c = threadIdx.x + q * kBlock;
      const int row = r0 + c / (W / 4);
      const int col = (c % (W / 4)) * 4;
      const bool is_a = col < MP;
      const int cc = is_a ? col : col - MP;
      const int width = is_a ? p.M : p.K;
      if constexpr (ALIGNED) {
        const bool live = (row < r_end) && (cc < width);
        const int rc = min(row, p.N - 1), ccc = min(cc, width - 4);
        const float4 v = *reinterpret_cast<const float4*>((is_a ? p.a : p.b) + (size_t)rc * width + ccc);
        stage[q] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < r_end) {
          const float* src = (is_a ? p.a : p.b) + (size_t)row * width + cc;
          if (cc < width) v.x = src[0];
          if (cc + 1 < width) v.y = src[1];
          if (cc + 2 < width) v.z = src[2];
          if (cc + 3 < width) v.w = src[3];
        }
        stage[q] = v;
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      tile4[buf][threadIdx.x + q * kBlock] = stage[q];
    }
  };
  auto multiply = [&](int buf) {
    const float* t = reinterpret_cast<const float*>(tile4[buf]);
    // operands of k-step kk+1 are read from LDS while the MFMAs of k-step kk issue
    float av[2][TM], bv[2][TK];
    auto read_ops = [&](int set, int kk) {
      const float* row = t + (2 * kk + half) * W;
#pragma unroll
      for (int i = 0; i < TM; ++i) av[set][i] = row[m_base + i * kTile + l31];
#pragma unroll
      for (int j = 0; j < TK; ++j) bv[set][j] = row[MP + k_base + j * kTile + l31];
    };
    read_ops(0, 0);
#pragma unroll
    for (int kk = 0; kk < kStageRows / 2; ++kk) {
      const int cur = kk & 1;
      if (kk + 1 < kStageRows / 2) read_ops(cur ^ 1, kk + 1);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        bsum[i] += av[cur][i];
#pragma unroll
        for (int j = 0; j < TK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (r_begin < r_end) {
    fetch(r_begin);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += kStageRows) {
      const bool more = r0 + kStageRows < r_end;
      if (more) fetch(r0 + kStageRows);        // in flight while this stage is multiplied
      multiply(buf);
      if (more) commit(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // partial of this slab: ws[block][m*K + k] and ws[block][M*K + m]
  float* out = p.ws + (size_t)blockIdx.x * p.out_cols;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int k = k_base + j * kTile + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_base + i * kTile + (r & 3) + 8 * (r >> 2) + 4 * half;    // C/D layout of 32x32 MFMA
        if (m < p.M && k < p.K) out[(size_t)m * p.K + k] = acc[i][j][r];
      }
    }
  if (wk == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float s = bsum[i] + __shfl_xor(bsum[i], 32);            // even + odd rows
      const int m = m_base + i * kTile + l31;
      if (half == 0 && m < p.M) out[(size_t)p.M * p.K + m] = s;
    }
  }
}

struct WgradPlan { int wm, wk, tm, tk; };

// smallest wave-layout x per-wave tiling that covers tiles_m x tiles_k with <= 8 tiles per wave
static bool plan_wgrad(int tiles_m, int tiles_k, WgradPlan* out) {
  static const int layouts[3][2] = {{2, 2}, {4, 1}, {1, 4}};
  static const int sizes[3] = {1, 2, 4};
  int best = 1 << 30;
  bool found = false;
  for (auto& lay : layouts)
    for (int tm : sizes)
      for (int tk : sizes) {
        if (tm * tk > 8) continue;
        if (lay[0] * tm < tiles_m || lay[1] * tk < tiles_k) continue;
        const int cost = (lay[0] * tm) * (lay[1] * tk) * 16 + (tm + tk);     // padded MFMA work, then loads
        if (cost < best) { best = cost; *out = {lay[0], lay[1], tm, tk}; found = true; }
      }
  return found;
}

static int wgrad_blocks(int64_t N) {
  int64_t b = (N + 511) / 512;            // at least 512 rows per slab
  if (b > 512) b = 512;                   // 2 workgroups per CU
  if (b < 1) b = 1;
  return (int)b;
}

#define MLGNN_WG_CASE(WM_, WK_, TM_, TK_)                                                            \
  if (pl.wm == WM_ && pl.wk == WK_ && pl.tm == TM_ && pl.tk == TK_) {                                \
    if (aligned) hipLaunchKernelGGL((linear_wgrad_kernel<WM_, WK_, TM_, TK_, true>), grid, block, 0, s, a);   \
    else hipLaunchKernelGGL((linear_wgrad_kernel<WM_, WK_, TM_, TK_, false>), grid, block, 0, s, a);        \
    launched = true;                                                                                 \
  }
#define MLGNN_WG_LAYOUT(WM_, WK_)                                                                    \
  MLGNN_WG_CASE(WM_, WK_, 1, 1) MLGNN_WG_CASE(WM_, WK_, 1, 2) MLGNN_WG_CASE(WM_, WK_, 1, 4)          \
  MLGNN_WG_CASE(WM_, WK_, 2, 1) MLGNN_WG_CASE(WM_, WK_, 2, 2) MLGNN_WG_CASE(WM_, WK_, 2, 4)          \
  MLGNN_WG_CASE(WM_, WK_, 4, 1) MLGNN_WG_CASE(WM_, WK_, 4, 2)

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_linear_wgrad_workspace_floats(int64_t N, int64_t M, int64_t K) {
  if (N < 0 || M <= 0 || K <= 0) return MLGNN_E_SHAPE;
  WgradPlan pl;
  if (!plan_wgrad((int)((M + kTile - 1) / kTile), (int)((K + kTile - 1) / kTile), &pl)) return MLGNN_E_SHAPE;
  return (int64_t)wgrad_blocks(N) * (M * K + M);
}

extern "C" int mlgnn_linear_wgrad(const void* grad_out, const void* x, float* grad_w_b, float* workspace,
                                  int64_t workspace_floats, int64_t N, int64_t M, int64_t K, int dtype,
                                  void* stream) {
  if (dtype != MLGNN_DTYPE_F32) return MLGNN_E_DTYPE;
  if (N < 0 || M <= 0 || K <= 0 || N > INT32_MAX || M * K > (1 << 24)) return MLGNN_E_SHAPE;
  WgradPlan pl;
  if (!plan_wgrad((int)((M + kTile - 1) / kTile), (int)((K + kTile - 1) / kTile), &pl)) return MLGNN_E_SHAPE;
  if (!grad_w_b || !workspace) return MLGNN_E_NULL;
  if (N > 0 && (!grad_out || !x)) return MLGNN_E_NULL;
  const int nblk = wgrad_blocks(N);
  const int cols = (int)(M * K + M);
  if (workspace_floats < (int64_t)nblk * cols) return MLGNN_E_WORKSPACE;
  WgradArgs a;
  a.a = (const float*)grad_out; a.b = (const float*)x; a.ws = workspace;
  a.N = (int)N; a.M = (int)M; a.K = (int)K; a.out_cols = cols;
  int rpb = (int)((N + nblk - 1) / nblk);
  rpb = (rpb + kStageRows - 1) / kStageRows * kStageRows;
  a.rows_per_block = rpb;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(nblk), block(kBlock);
  bool launched = false;
  const bool aligned = (M % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  MLGNN_WG_LAYOUT(2, 2) MLGNN_WG_LAYOUT(4, 1) MLGNN_WG_LAYOUT(1, 4)
  if (!launched) return MLGNN_E_SHAPE;
  int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_w_b, nblk, cols, s);
  return (int)hipGetLastError();
}
