// Weight / bias gradient of a Linear layer over a very tall activation matrix:
//     dW[M,K] = A^T B,   db[M] = column sums of A,   A = grad_out [N,M],  B = x [N,K],  N >> M,K
//
// Reference: autograd of nn.Linear inside MLP (models/gcn_lib/sparse/torch_nn.py:54-75, the dense
// epilogue of every conv) -- a "TN" GEMM whose reduction dimension is the 640 000 node rows and
// whose output is 128x256: the library picks a 32x32-tile kernel without split-K (~39 TFLOP/s
// measured).  Here the rows are split over the chip (one slab per workgroup), each wave keeps its
// share of the [M,K] output in fp32 MFMA accumulators (v_mfma_f32_32x32x2_f32: exact fp32 FMA
// chain), operands go straight from global memory into the MFMA operand registers (lane l of the
// A operand = grad_out[row + (l>>5)][m0 + (l&31)]: each half-wave reads one full 128-B line), and
// per-slab partials are summed in a fixed order by reduce_partials (bitwise reproducible).
//
// MFMA-bound: 2*N*M*K FLOP at the 157 TFLOP/s fp32-matrix peak; reads A and B once from HBM.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kTile = 32;
constexpr int kWgUnroll = 4;      // k-steps (2 rows each) whose operand loads are issued together

struct WgradArgs {
  const float* a; const float* b; float* ws;
  int N; int M; int K; int rows_per_block; int out_cols;   // out_cols = M*K + M
};

// wave layout WM x WK, tiles per wave TM x TK
template <int WM, int WK, int TM, int TK>
__global__ __launch_bounds__(kBlock) void linear_wgrad_kernel(const WgradArgs p) {
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

  bool a_ok[TM], b_ok[TK];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_ok[i] = (m_base + i * kTile + l31) < p.M;
#pragma unroll
  for (int j = 0; j < TK; ++j) b_ok[j] = (k_base + j * kTile + l31) < p.K;

  const int r_begin = blockIdx.x * p.rows_per_block;
  const int r_end = min(p.N, r_begin + p.rows_per_block);
  for (int r0 = r_begin; r0 < r_end; r0 += 2 * kWgUnroll) {
    float av[kWgUnroll][TM], bv[kWgUnroll][TK];
#pragma unroll
    for (int u = 0; u < kWgUnroll; ++u) {
      const int row = r0 + 2 * u + half;
      const bool rok = row < r_end;
      const float* ap = p.a + (size_t)row * p.M + m_base + l31;
      const float* bp = p.b + (size_t)row * p.K + k_base + l31;
#pragma unroll
      for (int i = 0; i < TM; ++i) av[u][i] = (rok && a_ok[i]) ? ap[i * kTile] : 0.f;
#pragma unroll
      for (int j = 0; j < TK; ++j) bv[u][j] = (rok && b_ok[j]) ? bp[j * kTile] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kWgUnroll; ++u) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        bsum[i] += av[u][i];
#pragma unroll
        for (int j = 0; j < TK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
      }
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
    hipLaunchKernelGGL((linear_wgrad_kernel<WM_, WK_, TM_, TK_>), grid, block, 0, s, a);             \
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
  rpb = (rpb + 2 * kWgUnroll - 1) / (2 * kWgUnroll) * (2 * kWgUnroll);
  a.rows_per_block = rpb;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(nblk), block(kBlock);
  bool launched = false;
  MLGNN_WG_LAYOUT(2, 2) MLGNN_WG_LAYOUT(4, 1) MLGNN_WG_LAYOUT(1, 4)
  if (!launched) return MLGNN_E_SHAPE;
  int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_w_b, nblk, cols, s);
  return (int)hipGetLastError();
}
