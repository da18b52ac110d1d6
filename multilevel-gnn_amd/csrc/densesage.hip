// DenseSAGEConv on a batch of small pooled graphs, one fused launch forward and one backward.
//
// Reference: torch_geometric.nn.DenseSAGEConv (PyG 2.2.0) as used by SAGEConvolutions / DiffPoolLayer
// (models/diff_pooling.py:24-32,45-53,58-65):
//     out = lin_rel(A x / clamp(rowsum A, 1)) + lin_root(x);   y = out / max(||out||_2, 1e-12)
// -- in the reference a batched matmul, a row sum, clamp, divide, two Linears, an add, and norm / clamp /
// divide for the normalisation (13 launches forward, about twice that backward, each latency bound on
// 146-, 37- or 10-node graphs).  Here ONE workgroup per pooled graph:
//   forward   P = x W_rel^T;  out = (A P) / deg + x W_root^T + b;  y = out * rinv        (rinv saved)
//             (A x) W^T = A (x W^T): the [n,C] neighbour mean never exists, only the [n,O] projection
//   backward  g = (gy - y <y, gy>) * rinv;   gP = A^T (g / deg);
//             gx = gP W_rel + g W_root;   gW_rel = gP^T x;   gW_root = g^T x;   gb = colsum g;
//             gA_ij = (<g_i, P_j> - [rowsum_i > 1] c_i) / deg_i,   c_i = sum_j A_ij <g_i, P_j> / deg_i
// Every product runs on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, 1e-4 parity) with operands read in
// MFMA layout from global memory or LDS; weight gradients leave as one partial per workgroup and are summed in
// a fixed order (reduce_partials).  Limits: n <= 160, C <= 128, O <= 64; the adjacency gradient (pooled
// levels below the first, where A itself was produced by DiffPool) additionally needs n <= 48.
#include "common.h"
#include "mlgnn.h"
#include "tile_gemm.h"

namespace mlgnn {

constexpr int kDsMaxN = 160, kDsMaxC = 128, kDsMaxO = 64, kDsMaxNAdj = 48;
// 16 waves per pooled graph: the products are chains of small latency-bound tiles, and the tiles of one product
// are independent -- more waves, not more work per wave, is what shortens the critical path
constexpr int kDsBlock = 1024, kDsWaves = kDsBlock / kWave;
constexpr int kDsSO = kDsMaxO + 1;            // odd LDS strides: row and transposed reads stay conflict free
constexpr float kDsNormEps = 1e-12f;          // F.normalize eps

struct DsArgs {                    // x, adj, weights, gy, y, y_out, gx, gadj: T (fp32 or bf16 storage); bias, rinv, ws: fp32
  const void* x; const void* adj; const void* w_rel; const void* w_root; const float* bias;
  const void* gy; const void* y; const float* rinv_in;
  void* y_out; float* rinv; void* gx; void* gadj; float* ws;
  int n; int C; int O; int adj_batched; int normalize; int ws_cols;
};

// row sums of the adjacency: one wavefront per row, lanes across the columns (coalesced), shuffle reduction
template <typename AB>
__device__ __forceinline__ void ds_degrees(const AB ab, int n, float* deg, float* raw) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  for (int r = wave; r < n; r += kDsWaves) {
    float s = 0.f;
    for (int j = lane; j < n; j += kWave) s += ab[(size_t)r * n + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
      if (raw) raw[r] = s;
      deg[r] = fmaxf(s, 1.0f);
    }
  }
}

// LDS images are sized by the ACTUAL (padded) widths of the call (dynamic shared memory, odd row strides so that row and
// transposed reads stay conflict free): a 32-channel conv needs 42 KB forward / 44 KB backward instead of the 125 / 106 KB
// of the largest supported shape, and two 16-wave workgroups then share a CU -- the 384 pooled graphs of a
// BASELINE configs[1] batch run as one resident round instead of two.
struct DsLds { int so, sx, np; };
__host__ __device__ inline DsLds ds_lds(int n, int C, int O) {
  DsLds l;
  l.np = (n + 15) & ~15;
  l.so = ((O + 15) & ~15) + 1;
  l.sx = C + 1 + (C & 1);                      // odd
  return l;
}
inline size_t ds_fwd_lds_bytes(int n, int C, int O) {
  const DsLds l = ds_lds(n, C, O);
  return ((size_t)l.np * l.so + (size_t)l.np * l.sx + l.np) * 4;
}
inline size_t ds_bwd_lds_bytes(int n, int C, int O, bool adj_grad) {
  const DsLds l = ds_lds(n, C, O);
  size_t f = (size_t)2 * l.np * l.so + 2 * l.np;
  if (adj_grad) f += (size_t)l.np * l.so + (size_t)l.np * (l.np + 1);
  return f * 4;
}

template <typename T>
__global__ __launch_bounds__(kDsBlock) void dense_sage_fwd_kernel(const DsArgs p) {
  extern __shared__ __attribute__((aligned(16))) float ds_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const int n = p.n, C = p.C, O = p.O;
  const DsLds L = ds_lds(n, C, O);
  const int SO = L.so, SX = L.sx;
  float* Pm_ = ds_smem;                        // [NP][SO]  x W_rel^T, then the un-normalised output
  float* Xm_ = Pm_ + (size_t)L.np * SO;        // [NP][SX]  the pooled graph's features, staged once (both products read them)
  float* deg = Xm_ + (size_t)L.np * SX;        // [NP]
#define P(r, c) Pm_[(r) * SO + (c)]
#define X(r, c) Xm_[(r) * SX + (c)]
  const StoredIn<T> xb{static_cast<const T*>(p.x) + (size_t)b * n * C};
  const StoredIn<T> ab{static_cast<const T*>(p.adj) + (p.adj_batched ? (size_t)b * n * n : 0)};
  const StoredIn<T> w_rel{static_cast<const T*>(p.w_rel)}, w_root{static_cast<const T*>(p.w_root)};
  const int NP = (n + 15) & ~15, OP = (O + 15) & ~15;
  const int Nt = NP / 16, Ot = OP / 16;
  const int l15 = lane & 15, lq = lane >> 4;

  ds_degrees(ab, n, deg, nullptr);
  for (int idx = tid; idx < NP * C; idx += kDsBlock) {               // coalesced; rows past n are zero
    const int r = idx / C, c = idx % C;
    X(r, c) = r < n ? xb[idx] : 0.f;
  }
  __syncthreads();
  // ---- P = x W_rel^T  [n, O] ----------------------------------------------------------------------
  for (int t = wave; t < Nt * Ot; t += kDsWaves) {
    const int i0 = (t / Ot) * 16, j0 = (t % Ot) * 16;
    const f32x4 acc = tile_gemm(C,
        [&](int i, int k) { return k < C ? X(i0 + i, k) : 0.f; },
        [&](int k, int j) { return (j0 + j < O && k < C) ? w_rel[(size_t)(j0 + j) * C + k] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) P(i0 + lq * 4 + r, j0 + l15) = acc[r];
  }
  __syncthreads();
  // ---- out = (A P) / deg + x W_root^T + b: every wave owns whole tiles, written back after a barrier -----
  f32x4 keep[(kDsMaxN / 16) * (kDsMaxO / 16) / kDsWaves + 1];
  int nk = 0;
  for (int t = wave; t < Nt * Ot; t += kDsWaves, ++nk) {
    const int i0 = (t / Ot) * 16, j0 = (t % Ot) * 16;
    f32x4 acc = tile_gemm(n,
        [&](int i, int k) { return (i0 + i < n && k < n) ? ab[(size_t)(i0 + i) * n + k] : 0.f; },
        [&](int k, int j) { return k < n ? P(k, j0 + j) : 0.f; });
    const f32x4 root = tile_gemm(C,
        [&](int i, int k) { return k < C ? X(i0 + i, k) : 0.f; },
        [&](int k, int j) { return (j0 + j < O && k < C) ? w_root[(size_t)(j0 + j) * C + k] : 0.f; });
    const float bj = (p.bias && j0 + l15 < O) ? p.bias[j0 + l15] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r;
      acc[r] = acc[r] / deg[min(row, n - 1)] + root[r] + bj;
    }
    keep[nk] = acc;
  }
  __syncthreads();                               // every read of P as the projection is done
  nk = 0;
  for (int t = wave; t < Nt * Ot; t += kDsWaves, ++nk) {
    const int i0 = (t / Ot) * 16, j0 = (t % Ot) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) P(i0 + lq * 4 + r, j0 + l15) = keep[nk][r];
  }
  __syncthreads();
  // ---- row normalisation and store ----------------------------------------------------------------
  for (int r = wave; r < n; r += kDsWaves) {
    const float v = lane < O ? P(r, lane) : 0.f;              // O <= 64: one lane per output channel
    float ss = v * v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float ri = p.normalize ? 1.0f / fmaxf(sqrtf(ss), kDsNormEps) : 1.0f;
    if (lane < O) stored_write<T>(p.y_out, ((size_t)b * n + r) * O + lane, v * ri);
    if (lane == 0) p.rinv[(size_t)b * n + r] = ri;
  }
}
#undef P
#undef X

template <typename T>
__global__ __launch_bounds__(kDsBlock) void dense_sage_bwd_kernel(const DsArgs p) {
  extern __shared__ __attribute__((aligned(16))) float ds_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const int n = p.n, C = p.C, O = p.O;
  const DsLds L = ds_lds(n, C, O);
  const int SO = L.so, SM = L.np + 1;
  float* G_ = ds_smem;                          // [NP][SO]  g = d loss / d out
  float* GP_ = G_ + (size_t)L.np * SO;          // [NP][SO]  gP = A^T (g / deg)
  float* deg = GP_ + (size_t)L.np * SO;         // [NP]
  float* raw = deg + L.np;                      // [NP]
  float* Pm_ = raw + L.np;                      // [NP][SO]  P = x W_rel^T        (adjacency gradient only)
  float* M_ = Pm_ + (size_t)L.np * SO;          // [NP][NP+1] <g_i, P_j>          (adjacency gradient only)
#define G(r, c) G_[(r) * SO + (c)]
#define GP(r, c) GP_[(r) * SO + (c)]
#define Pm(r, c) Pm_[(r) * SO + (c)]
#define M(r, c) M_[(r) * SM + (c)]
  const StoredIn<T> xb{static_cast<const T*>(p.x) + (size_t)b * n * C};
  const StoredIn<T> ab{static_cast<const T*>(p.adj) + (p.adj_batched ? (size_t)b * n * n : 0)};
  const StoredIn<T> w_rel{static_cast<const T*>(p.w_rel)}, w_root{static_cast<const T*>(p.w_root)};
  const StoredIn<T> gyb{static_cast<const T*>(p.gy) + (size_t)b * n * O};
  const StoredIn<T> yb{static_cast<const T*>(p.y) + (size_t)b * n * O};
  const int NP = (n + 15) & ~15, OP = (O + 15) & ~15, CP = (C + 15) & ~15;
  const int Nt = NP / 16, Ot = OP / 16, Ct = CP / 16;
  const int l15 = lane & 15, lq = lane >> 4;

  ds_degrees(ab, n, deg, raw);
  // ---- g = (gy - y <y, gy>) * rinv, zero padded ------------------------------------------------------
  for (int r = wave; r < NP; r += kDsWaves) {
    float gv = 0.f, yv = 0.f;
    if (r < n && lane < O) { gv = gyb[(size_t)r * O + lane]; yv = yb[(size_t)r * O + lane]; }
    float dot = gv * yv;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    const float ri = r < n ? p.rinv_in[(size_t)b * n + r] : 0.f;
    const float g = p.normalize ? (gv - yv * dot) * ri : gv;
    if (lane < OP) G(r, lane) = (r < n && lane < O) ? g : 0.f;
  }
  __syncthreads();
  // ---- gP = A^T (g / deg)  [n, O] ---------------------------------------------------------------------
  for (int t = wave; t < Nt * Ot; t += kDsWaves) {
    const int i0 = (t / Ot) * 16, j0 = (t % Ot) * 16;
    const f32x4 acc = tile_gemm(n,
        [&](int i, int k) { return (i0 + i < n && k < n) ? ab[(size_t)k * n + i0 + i] : 0.f; },
        [&](int k, int j) { return k < n ? G(k, j0 + j) / deg[k] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) GP(i0 + lq * 4 + r, j0 + l15) = acc[r];
  }
  __syncthreads();
  // ---- gx = gP W_rel + g W_root  [n, C] ------------------------------------------------------------------
  const size_t gx_off = (size_t)b * n * C;
  for (int t = wave; t < Nt * Ct; t += kDsWaves) {
    const int i0 = (t / Ct) * 16, c0 = (t % Ct) * 16;
    f32x4 acc = tile_gemm(O,
        [&](int i, int k) { return k < O ? GP(i0 + i, k) : 0.f; },
        [&](int k, int j) { return (k < O && c0 + j < C) ? w_rel[(size_t)k * C + c0 + j] : 0.f; });
    const f32x4 t2 = tile_gemm(O,
        [&](int i, int k) { return k < O ? G(i0 + i, k) : 0.f; },
        [&](int k, int j) { return (k < O && c0 + j < C) ? w_root[(size_t)k * C + c0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lq * 4 + r, col = c0 + l15;
      if (row < n && col < C) stored_write<T>(p.gx, gx_off + (size_t)row * C + col, acc[r] + t2[r]);
    }
  }
  // ---- weight / bias gradient partials: ws[b] = [gW_rel (O*C) | gW_root (O*C) | gb (O)] ---------------------
  float* wsb = p.ws + (size_t)b * p.ws_cols;
  for (int t = wave; t < 2 * Ot * Ct; t += kDsWaves) {
    const int which = t / (Ot * Ct), tt = t % (Ot * Ct);
    const int o0 = (tt / Ct) * 16, c0 = (tt % Ct) * 16;
    const float* src = which == 0 ? GP_ : G_;
    const f32x4 acc = tile_gemm(n,
        [&](int i, int k) { return k < n ? src[k * SO + o0 + i] : 0.f; },
        [&](int k, int j) { return (k < n && c0 + j < C) ? xb[(size_t)k * C + c0 + j] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = o0 + lq * 4 + r, c = c0 + l15;
      if (o < O && c < C) wsb[(size_t)which * O * C + (size_t)o * C + c] = acc[r];
    }
  }
  for (int o = tid; o < O; o += kDsBlock) {
    float s = 0.f;
    for (int r = 0; r < n; ++r) s += G(r, o);
    wsb[(size_t)2 * O * C + o] = s;
  }
  // ---- adjacency gradient (n <= 48) ---------------------------------------------------------------------------
  if (p.gadj) {
    for (int t = wave; t < Nt * Ot; t += kDsWaves) {            // P = x W_rel^T
      const int i0 = (t / Ot) * 16, j0 = (t % Ot) * 16;
      const f32x4 acc = tile_gemm(C,
          [&](int i, int k) { return (i0 + i < n && k < C) ? xb[(size_t)(i0 + i) * C + k] : 0.f; },
          [&](int k, int j) { return (j0 + j < O && k < C) ? w_rel[(size_t)(j0 + j) * C + k] : 0.f; });
#pragma unroll
      for (int r = 0; r < 4; ++r) Pm(i0 + lq * 4 + r, j0 + l15) = acc[r];
    }
    __syncthreads();
    for (int t = wave; t < Nt * Nt; t += kDsWaves) {            // M = g P^T
      const int i0 = (t / Nt) * 16, j0 = (t % Nt) * 16;
      const f32x4 acc = tile_gemm(O,
          [&](int i, int k) { return k < O ? G(i0 + i, k) : 0.f; },
          [&](int k, int j) { return k < O ? Pm(j0 + j, k) : 0.f; });
#pragma unroll
      for (int r = 0; r < 4; ++r) M(i0 + lq * 4 + r, j0 + l15) = acc[r];
    }
    __syncthreads();
    const size_t ga_off = (size_t)b * n * n;
    for (int r = wave; r < n; r += kDsWaves) {
      float c = lane < n ? ab[(size_t)r * n + lane] * M(r, lane) : 0.f;      // n <= 48 < 64 lanes
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
      const float ci = raw[r] > 1.0f ? c / deg[r] : 0.f;                     // clamp(rowsum, 1) passes gradient above 1
      if (lane < n) stored_write<T>(p.gadj, ga_off + (size_t)r * n + lane, (M(r, lane) - ci) / deg[r]);
    }
  }
}
#undef G
#undef GP
#undef Pm
#undef M

static bool ds_dims_ok(int64_t n, int64_t C, int64_t O) {
  return n >= 1 && n <= kDsMaxN && C >= 1 && C <= kDsMaxC && O >= 1 && O <= kDsMaxO;
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_dense_sage_supported(int64_t n, int64_t C, int64_t O, int need_grad_adj) {
  return (ds_dims_ok(n, C, O) && (!need_grad_adj || n <= kDsMaxNAdj)) ? 1 : 0;
}

extern "C" int64_t mlgnn_dense_sage_bwd_workspace_floats(int64_t B, int64_t C, int64_t O) {
  if (B < 0 || C <= 0 || O <= 0) return MLGNN_E_SHAPE;
  return B * (2 * O * C + O);
}

extern "C" int mlgnn_dense_sage_fwd(const void* x, const void* adj, const void* w_rel, const void* w_root,
                                    const float* bias, void* y, float* rinv, int64_t B, int64_t n, int64_t C,
                                    int64_t O, int adj_batched, int normalize, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (B < 0 || !ds_dims_ok(n, C, O)) return MLGNN_E_SHAPE;
  if (B == 0) return 0;
  if (!x || !adj || !w_rel || !w_root || !y || !rinv) return MLGNN_E_NULL;
  DsArgs a = {};
  a.x = x; a.adj = adj; a.w_rel = w_rel; a.w_root = w_root;
  a.bias = bias; a.y_out = y; a.rinv = rinv;
  a.n = (int)n; a.C = (int)C; a.O = (int)O; a.adj_batched = adj_batched; a.normalize = normalize;
  const size_t lds = ds_fwd_lds_bytes((int)n, (int)C, (int)O);
  static size_t fwd_attr = 0;                   // (idempotent: a race only repeats the call)
  if (lds > fwd_attr) {
    const int most = (int)ds_fwd_lds_bytes(kDsMaxN, kDsMaxC, kDsMaxO);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_sage_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_sage_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    fwd_attr = (size_t)most;
  }
  if (dtype == MLGNN_DTYPE_BF16) hipLaunchKernelGGL(dense_sage_fwd_kernel<bf16_t>, dim3((unsigned)B), dim3(kDsBlock), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(dense_sage_fwd_kernel<float>, dim3((unsigned)B), dim3(kDsBlock), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_dense_sage_bwd(const void* grad_y, const void* y, const float* rinv, const void* x,
                                    const void* adj, const void* w_rel, const void* w_root, void* grad_x,
                                    void* grad_adj, float* grad_w, float* workspace, int64_t workspace_floats,
                                    int64_t B, int64_t n, int64_t C, int64_t O, int adj_batched, int normalize,
                                    int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (B < 0 || !ds_dims_ok(n, C, O)) return MLGNN_E_SHAPE;
  if (grad_adj && n > kDsMaxNAdj) return MLGNN_E_SHAPE;
  if (!grad_w) return MLGNN_E_NULL;
  const int cols = (int)(2 * O * C + O);
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) return (int)hipMemsetAsync(grad_w, 0, (size_t)cols * 4, s);
  if (!grad_y || !y || !rinv || !x || !adj || !w_rel || !w_root || !grad_x || !workspace) return MLGNN_E_NULL;
  if (workspace_floats < B * cols) return MLGNN_E_WORKSPACE;
  DsArgs a = {};
  a.x = x; a.adj = adj; a.w_rel = w_rel; a.w_root = w_root;
  a.gy = grad_y; a.y = y; a.rinv_in = rinv;
  a.gx = grad_x; a.gadj = grad_adj; a.ws = workspace; a.ws_cols = cols;
  a.n = (int)n; a.C = (int)C; a.O = (int)O; a.adj_batched = adj_batched; a.normalize = normalize;
  const size_t lds = ds_bwd_lds_bytes((int)n, (int)C, (int)O, grad_adj != nullptr);
  static size_t bwd_attr = 0;
  if (lds > bwd_attr) {
    const size_t most = ds_bwd_lds_bytes(kDsMaxN, kDsMaxC, kDsMaxO, false) > ds_bwd_lds_bytes(kDsMaxNAdj, kDsMaxC, kDsMaxO, true)
                            ? ds_bwd_lds_bytes(kDsMaxN, kDsMaxC, kDsMaxO, false) : ds_bwd_lds_bytes(kDsMaxNAdj, kDsMaxC, kDsMaxO, true);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_sage_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_sage_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
    bwd_attr = most;
  }
  if (dtype == MLGNN_DTYPE_BF16) hipLaunchKernelGGL(dense_sage_bwd_kernel<bf16_t>, dim3((unsigned)B), dim3(kDsBlock), lds, s, a);
  else hipLaunchKernelGGL(dense_sage_bwd_kernel<float>, dim3((unsigned)B), dim3(kDsBlock), lds, s, a);
  int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_w, (int)B, cols, s);
  return (int)hipGetLastError();
}
