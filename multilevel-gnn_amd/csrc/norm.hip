// Fused LayerNorm (+ ReLU) forward / backward over [rows, d]: fp32 (d <= 256 with d % 4 == 0, or d <= 512 with
// d % 8 == 0) or bf16 storage with fp32 arithmetic (d <= 512, d % 8 == 0).
//
// Reference: norm_layer('layer') + act_layer('relu') as chained by MLP
// (models/gcn_lib/sparse/torch_nn.py:27-38,54-75) and by the res+ block
// (models/deepergcn.py:236-241: norms[l-1](h) -> relu).  ATen runs LayerNorm and ReLU as separate
// passes forward and three kernels backward; here one pass each way.
//
// Lane layout as in the aggregation kernels: LPR = next_pow2(d/VEC) lanes hold one row (16 bytes per
// lane), a wave works on 64/LPR rows at once, row statistics are xor-shuffle reductions inside
// the lane group.  HBM-bound: forward 2*rows*d*4 bytes, backward 3*rows*d*4 bytes (+ 8 B/row stats).
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

template <int LPR_LOG2>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int off = 1; off < (1 << LPR_LOG2); off <<= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

template <int LPR_LOG2>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = 1; off < (1 << LPR_LOG2); off <<= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int kLnRows = 4;            // row groups in flight per wave

struct LnArgs {
  const void* x; const void* go; const float* gamma; const float* beta; const void* gextra;
  void* out; float* mean; float* rstd; void* gx; float* ws; float* rowmax;
  const uint8_t* keep; float keep_scale;        // dropout behind the activation: out *= keep ? keep_scale : 0
  int rows; int d; float eps; int relu;
};

// VEC one-byte keep flags -> multipliers (0 or scale)
template <int VEC>
__device__ __forceinline__ void load_keep(float (&m)[VEC], const uint8_t* p, float scale) {
  if constexpr (VEC == 4) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = ((w >> (8 * i)) & 0xffu) ? scale : 0.f;
  } else {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[i] = ((w.x >> (8 * i)) & 0xffu) ? scale : 0.f;
      m[4 + i] = ((w.y >> (8 * i)) & 0xffu) ? scale : 0.f;
    }
  }
}

// T = float (VEC 4) or bf16_t (VEC 8): 16 bytes per lane either way; statistics and arithmetic in fp32
template <typename T, int VEC, int LPR_LOG2>
__global__ __launch_bounds__(kBlock) void layernorm_act_fwd_kernel(const LnArgs a) {
  constexpr int LPR = 1 << LPR_LOG2, GROUPS = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / LPR, cl = lane % LPR, c0 = cl * VEC;
  const bool cact = c0 < a.d;
  const int wave_global = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  const T* x = static_cast<const T*>(a.x);
  T* out = static_cast<T*>(a.out);
  float g[VEC], b[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { g[i] = 0.f; b[i] = 0.f; }
  if (cact) { load_vec<VEC>(g, a.gamma + c0); load_vec<VEC>(b, a.beta + c0); }
  const float inv_d = 1.0f / (float)a.d;
  // kLnRows row groups per wave and iteration: their loads are issued together (the reductions that follow
  // are dependent chains; one row at a time leaves the memory pipe idle behind them)
  for (int r0 = wave_global * GROUPS * kLnRows; r0 < a.rows; r0 += n_waves * GROUPS * kLnRows) {
    float v[kLnRows][VEC];
    bool ok[kLnRows];
#pragma unroll
    for (int u = 0; u < kLnRows; ++u) {
      const int r = r0 + u * GROUPS + sub;
      ok[u] = (r < a.rows) && cact;
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
      if (ok[u]) load_t<T, VEC>(v[u], x + (size_t)r * a.d + c0);
    }
#pragma unroll
    for (int u = 0; u < kLnRows; ++u) {
      const int r = r0 + u * GROUPS + sub;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) sum += v[u][i];
      const float mu = group_sum<LPR_LOG2>(sum) * inv_d;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float c = cact ? v[u][i] - mu : 0.f; q = fmaf(c, c, q); }
      const float rs = rsqrtf(group_sum<LPR_LOG2>(q) * inv_d + a.eps);
      float om = 0.f;
      if (ok[u]) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float y = fmaf((v[u][i] - mu) * rs, g[i], b[i]);
          o[i] = a.relu ? relu_keep_nan(y) : y;
        }
        if (a.keep) {
          float km[VEC];
          load_keep<VEC>(km, a.keep + (size_t)r * a.d + c0, a.keep_scale);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] *= km[i];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) om = fmaxf(om, fabsf(o[i]));
        store_t<T, VEC>(out + (size_t)r * a.d + c0, o);
        if (cl == 0) { a.mean[r] = mu; a.rstd[r] = rs; }
      }
      if (a.rowmax) {                          // max |row| for the consumer GEMM's per-row scaling (tallgemm.hip)
        om = group_max<LPR_LOG2>(om);
        if (ok[u] && cl == 0) a.rowmax[r] = om;
      }
    }
  }
}

template <typename T, int VEC, int LPR_LOG2>
__global__ __launch_bounds__(kBlock) void layernorm_act_bwd_kernel(const LnArgs a) {
  constexpr int LPR = 1 << LPR_LOG2, GROUPS = kWave / LPR;
  __shared__ float red[kWavesPerBlock][2][kWave * VEC];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane / LPR, cl = lane % LPR, c0 = cl * VEC;
  const bool cact = c0 < a.d;
  const int wave_global = blockIdx.x * kWavesPerBlock + wave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  const T* x = static_cast<const T*>(a.x);
  const T* gout = static_cast<const T*>(a.go);
  const T* gextra = static_cast<const T*>(a.gextra);
  T* gx = static_cast<T*>(a.gx);
  float g[VEC], b[VEC], dg[VEC], db[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { g[i] = 0.f; b[i] = 0.f; dg[i] = 0.f; db[i] = 0.f; }
  if (cact) { load_vec<VEC>(g, a.gamma + c0); load_vec<VEC>(b, a.beta + c0); }
  const float inv_d = 1.0f / (float)a.d;
  for (int r0 = wave_global * GROUPS * kLnRows; r0 < a.rows; r0 += n_waves * GROUPS * kLnRows) {
    float v[kLnRows][VEC], go[kLnRows][VEC], mu[kLnRows], rs[kLnRows], nsc[kLnRows];
    bool ok[kLnRows];
#pragma unroll
    for (int u = 0; u < kLnRows; ++u) {
      const int r = r0 + u * GROUPS + sub;
      ok[u] = (r < a.rows) && cact;
      mu[u] = 0.f; rs[u] = 0.f; nsc[u] = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) { v[u][i] = 0.f; go[u][i] = 0.f; }
      if (ok[u]) {
        load_t<T, VEC>(v[u], x + (size_t)r * a.d + c0);
        load_t<T, VEC>(go[u], gout + (size_t)r * a.d + c0);
        if (a.keep) {
          float km[VEC];
          load_keep<VEC>(km, a.keep + (size_t)r * a.d + c0, a.keep_scale);
#pragma unroll
          for (int i = 0; i < VEC; ++i) go[u][i] *= km[i];
        }
        mu[u] = a.mean ? a.mean[r] : 0.f; rs[u] = a.rstd[r]; nsc[u] = a.mean ? rs[u] : 1.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < kLnRows; ++u) {
      const int r = r0 + u * GROUPS + sub;
      float xh[VEC], gg[VEC], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        xh[i] = (v[u][i] - mu[u]) * nsc[u];       // nsc = rstd, or 1 with mu = 0 when x is already normalised
        const float y = fmaf(xh[i], g[i], b[i]);
        const float gy = (a.relu && !(y > 0.f)) ? 0.f : go[u][i];
        dg[i] = fmaf(gy, xh[i], dg[i]);
        db[i] += gy;
        gg[i] = gy * g[i];
        s1 += gg[i];
        s2 = fmaf(gg[i], xh[i], s2);
      }
      s1 = group_sum<LPR_LOG2>(s1) * inv_d;
      s2 = group_sum<LPR_LOG2>(s2) * inv_d;
      float om = 0.f;
      if (ok[u]) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = rs[u] * (gg[i] - s1 - xh[i] * s2);
        if (gextra) {                        // gradient arriving at x on the block's identity branch
          float e[VEC];
          load_t<T, VEC>(e, gextra + (size_t)r * a.d + c0);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] += e[i];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) om = fmaxf(om, fabsf(o[i]));
        store_t<T, VEC>(gx + (size_t)r * a.d + c0, o);
      }
      if (a.rowmax) {
        om = group_max<LPR_LOG2>(om);
        if (ok[u] && cl == 0) a.rowmax[r] = om;
      }
    }
  }
  // d gamma / d beta: lane groups -> waves -> one [2,d] partial per workgroup
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1)
#pragma unroll
    for (int i = 0; i < VEC; ++i) { dg[i] += __shfl_xor(dg[i], off); db[i] += __shfl_xor(db[i], off); }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[wave][0][c0 + i] = dg[i]; red[wave][1][c0 + i] = db[i]; }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * a.d; idx += kBlock) {
    const int which = idx / a.d, c = idx % a.d;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) s += red[w][which][c];
    a.ws[((size_t)blockIdx.x * 2 + which) * a.d + c] = s;
  }
}

static int ln_grid(int64_t rows, int lpr_log2) {
  const int groups = kWave >> lpr_log2;
  const int64_t per_block = (int64_t)groups * kWavesPerBlock * kLnRows;
  int64_t blocks = (rows + per_block - 1) / per_block;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

static bool ln_ok(int64_t d) { return d > 0 && d <= 256 && d % 4 == 0; }
// channels per lane: bf16 8 (16 bytes); fp32 4 (16 bytes) up to d = 256, 8 (two 16-byte loads) for 256 < d <= 512
static int ln_vec(int dtype, int64_t d) { return (dtype == MLGNN_DTYPE_BF16 || d > 256) ? 8 : 4; }
// one row per wave at most: fp32 d <= 512 (d % 4 == 0 up to 256, d % 8 == 0 beyond); bf16 d <= 512, d % 8 == 0
static bool ln_ok_t(int64_t d, int dtype) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return false;
  const int v = ln_vec(dtype, d);
  return d > 0 && d <= 64 * v && d % v == 0;
}
static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define MLGNN_LN_LAUNCH(KERNEL, lpr, ...)                                         \
  switch (lpr) {                                                                  \
    case 0: hipLaunchKernelGGL((KERNEL<0>), __VA_ARGS__); break;                  \
    case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break;                  \
    case 2: hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__); break;                  \
    case 3: hipLaunchKernelGGL((KERNEL<3>), __VA_ARGS__); break;                  \
    case 4: hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__); break;                  \
    case 5: hipLaunchKernelGGL((KERNEL<5>), __VA_ARGS__); break;                  \
    default: hipLaunchKernelGGL((KERNEL<6>), __VA_ARGS__); break;                 \
  }

#define MLGNN_LNT_LAUNCH(KERNEL, T, VEC, lpr, ...)                                \
  switch (lpr) {                                                                  \
    case 0: hipLaunchKernelGGL((KERNEL<T, VEC, 0>), __VA_ARGS__); break;          \
    case 1: hipLaunchKernelGGL((KERNEL<T, VEC, 1>), __VA_ARGS__); break;          \
    case 2: hipLaunchKernelGGL((KERNEL<T, VEC, 2>), __VA_ARGS__); break;          \
    case 3: hipLaunchKernelGGL((KERNEL<T, VEC, 3>), __VA_ARGS__); break;          \
    case 4: hipLaunchKernelGGL((KERNEL<T, VEC, 4>), __VA_ARGS__); break;          \
    case 5: hipLaunchKernelGGL((KERNEL<T, VEC, 5>), __VA_ARGS__); break;          \
    default: hipLaunchKernelGGL((KERNEL<T, VEC, 6>), __VA_ARGS__); break;         \
  }

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_layernorm_bwd_workspace_floats(int64_t rows, int64_t d, int dtype) {
  if (rows < 0 || !ln_ok_t(d, dtype)) return MLGNN_E_SHAPE;
  return (int64_t)ln_grid(rows, lanes_per_row_log2(d, ln_vec(dtype, d))) * 2 * d;
}

extern "C" int mlgnn_layernorm_act_fwd(const void* x, const float* gamma, const float* beta, void* out,
                                       float* mean, float* rstd, float* row_max, const uint8_t* keep_mask,
                                       float keep_scale, int64_t rows, int64_t d, float eps,
                                       int relu, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (rows < 0 || rows > INT32_MAX || !ln_ok_t(d, dtype)) return MLGNN_E_SHAPE;
  if (rows == 0) return 0;
  if (!x || !gamma || !beta || !out || !mean || !rstd) return MLGNN_E_NULL;
  if (!a16(x) || !a16(out) || !a16(gamma) || !a16(beta)) return MLGNN_E_ALIGN;
  LnArgs a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.out = out; a.mean = mean; a.rstd = rstd;
  a.rowmax = row_max; a.keep = keep_mask; a.keep_scale = keep_scale;
  if (keep_mask && d % ln_vec(dtype, d) != 0) return MLGNN_E_SHAPE;
  if (keep_mask && (reinterpret_cast<uintptr_t>(keep_mask) % 8) != 0) return MLGNN_E_ALIGN;    // read 4 / 8 flags at a time
  a.rows = (int)rows; a.d = (int)d; a.eps = eps; a.relu = relu;
  const int lpr = lanes_per_row_log2(d, ln_vec(dtype, d));
  const dim3 grid(ln_grid(rows, lpr)), block(kBlock);
  if (dtype == MLGNN_DTYPE_F32 && d > 256) {
    MLGNN_LNT_LAUNCH(layernorm_act_fwd_kernel, float, 8, lpr, grid, block, 0, (hipStream_t)stream, a)
  } else if (dtype == MLGNN_DTYPE_F32) {
    MLGNN_LNT_LAUNCH(layernorm_act_fwd_kernel, float, 4, lpr, grid, block, 0, (hipStream_t)stream, a)
  } else {
    MLGNN_LNT_LAUNCH(layernorm_act_fwd_kernel, bf16_t, 8, lpr, grid, block, 0, (hipStream_t)stream, a)
  }
  return (int)hipGetLastError();
}

extern "C" int mlgnn_layernorm_act_bwd(const void* grad_out, const void* x, const float* gamma,
                                       const float* beta, const float* mean, const float* rstd,
                                       const void* grad_extra, void* grad_x, float* row_max, float* grad_gamma_beta, float* workspace,
                                       int64_t workspace_floats, const uint8_t* keep_mask, float keep_scale,
                                       int64_t rows, int64_t d, int relu,
                                       int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (rows < 0 || rows > INT32_MAX || !ln_ok_t(d, dtype)) return MLGNN_E_SHAPE;
  if (!grad_gamma_beta || !workspace) return MLGNN_E_NULL;
  const int lpr = lanes_per_row_log2(d, ln_vec(dtype, d));
  const int nblk = ln_grid(rows, lpr);
  if (workspace_floats < (int64_t)nblk * 2 * d) return MLGNN_E_WORKSPACE;
  if (rows > 0 && (!grad_out || !x || !gamma || !beta || !rstd || !grad_x)) return MLGNN_E_NULL;
  if (!a16(x) || !a16(grad_out) || !a16(grad_x) || !a16(gamma) || !a16(beta) || !a16(grad_extra)) return MLGNN_E_ALIGN;
  if (keep_mask && (reinterpret_cast<uintptr_t>(keep_mask) % 8) != 0) return MLGNN_E_ALIGN;
  LnArgs a{};
  a.x = x; a.go = grad_out; a.gamma = gamma; a.beta = beta;
  a.gextra = grad_extra; a.keep = keep_mask; a.keep_scale = keep_scale;
  a.mean = (float*)mean; a.rstd = (float*)rstd; a.gx = grad_x; a.ws = workspace; a.rowmax = row_max;
  a.rows = (int)rows; a.d = (int)d; a.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(nblk), block(kBlock);
  if (dtype == MLGNN_DTYPE_F32 && d > 256) {
    MLGNN_LNT_LAUNCH(layernorm_act_bwd_kernel, float, 8, lpr, grid, block, 0, s, a)
  } else if (dtype == MLGNN_DTYPE_F32) {
    MLGNN_LNT_LAUNCH(layernorm_act_bwd_kernel, float, 4, lpr, grid, block, 0, s, a)
  } else {
    MLGNN_LNT_LAUNCH(layernorm_act_bwd_kernel, bf16_t, 8, lpr, grid, block, 0, s, a)
  }
  int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_gamma_beta, nblk, 2 * (int)d, s);
  return (int)hipGetLastError();
}

// ================================================================================================
// MsgNorm fused with the root add:  h = x + normalize(m, p=2, dim=1) * ||x||_2 * scale
// Reference: MsgNorm.forward (models/gcn_lib/sparse/torch_message.py:175-179) followed by
// h = x + m (models/gcn_lib/sparse/torch_vertex.py:86-89).  F.normalize clamps the norm at 1e-12.
// Same lane layout as LayerNorm above; d <= 256, d % 4 == 0.
// ================================================================================================
namespace mlgnn {

constexpr float kNormEps = 1e-12f;

struct MnArgs {            // x / m / gh / h / gx / gm are T (fp32 or bf16); scale, ws fp32
  const void* x; const void* m; const void* gh; const float* scale;
  void* h; void* gx; void* gm; float* ws;
  int rows; int d;
};

template <typename T, int VEC, int LPR_LOG2>
__global__ __launch_bounds__(kBlock) void msgnorm_add_fwd_kernel(const MnArgs a) {
  constexpr int LPR = 1 << LPR_LOG2, GROUPS = kWave / LPR;
  const T* X = static_cast<const T*>(a.x);
  const T* M = static_cast<const T*>(a.m);
  T* H = static_cast<T*>(a.h);
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / LPR, cl = lane % LPR, c0 = cl * VEC;
  const bool cact = c0 < a.d;
  const int wave_global = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  const float s = a.scale[0];
  for (int r0 = wave_global * GROUPS; r0 < a.rows; r0 += n_waves * GROUPS) {
    const int r = r0 + sub;
    const bool ok = (r < a.rows) && cact;
    float xv[VEC], mv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { xv[i] = 0.f; mv[i] = 0.f; }
    if (ok) { load_t<T, VEC>(xv, X + (size_t)r * a.d + c0); load_t<T, VEC>(mv, M + (size_t)r * a.d + c0); }
    float qx = 0.f, qm = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { qx = fmaf(xv[i], xv[i], qx); qm = fmaf(mv[i], mv[i], qm); }
    const float nx = sqrtf(group_sum<LPR_LOG2>(qx));
    const float nm = fmaxf(sqrtf(group_sum<LPR_LOG2>(qm)), kNormEps);
    const float c = s * nx / nm;
    if (ok) {
      float o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = fmaf(mv[i], c, xv[i]);
      store_t<T, VEC>(H + (size_t)r * a.d + c0, o);
    }
  }
}

// gm = c*g - (c/nm^2)(g.m) m ;  gx = g + (g.m) s/(nm nx) x ;  gs = sum_rows (g.m) nx/nm   (g = grad of h)
template <typename T, int VEC, int LPR_LOG2>
__global__ __launch_bounds__(kBlock) void msgnorm_add_bwd_kernel(const MnArgs a) {
  constexpr int LPR = 1 << LPR_LOG2, GROUPS = kWave / LPR;
  __shared__ float red[kWavesPerBlock];
  const T* X = static_cast<const T*>(a.x);
  const T* M = static_cast<const T*>(a.m);
  const T* GH = static_cast<const T*>(a.gh);
  T* GX = static_cast<T*>(a.gx);
  T* GM = static_cast<T*>(a.gm);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane / LPR, cl = lane % LPR, c0 = cl * VEC;
  const bool cact = c0 < a.d;
  const int wave_global = blockIdx.x * kWavesPerBlock + wave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  const float s = a.scale[0];
  float gs = 0.f;
  for (int r0 = wave_global * GROUPS; r0 < a.rows; r0 += n_waves * GROUPS) {
    const int r = r0 + sub;
    const bool ok = (r < a.rows) && cact;
    float xv[VEC], mv[VEC], g[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { xv[i] = 0.f; mv[i] = 0.f; g[i] = 0.f; }
    if (ok) {
      load_t<T, VEC>(xv, X + (size_t)r * a.d + c0);
      load_t<T, VEC>(mv, M + (size_t)r * a.d + c0);
      load_t<T, VEC>(g, GH + (size_t)r * a.d + c0);
    }
    float qx = 0.f, qm = 0.f, gd = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { qx = fmaf(xv[i], xv[i], qx); qm = fmaf(mv[i], mv[i], qm); gd = fmaf(g[i], mv[i], gd); }
    const float nx = sqrtf(group_sum<LPR_LOG2>(qx));
    const float nm_raw = sqrtf(group_sum<LPR_LOG2>(qm));
    const float nm = fmaxf(nm_raw, kNormEps);
    gd = group_sum<LPR_LOG2>(gd);
    const float c = s * nx / nm;
    const float km = (nm_raw > kNormEps) ? c / (nm * nm) * gd : 0.f;       // clamp branch: norm is constant
    const float kx = (nx > 0.f) ? gd * s / (nm * nx) : 0.f;               // torch: d||x||/dx = 0 at x = 0
    if (ok) {
      float om[VEC], ox[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) { om[i] = fmaf(g[i], c, -km * mv[i]); ox[i] = fmaf(kx, xv[i], g[i]); }
      store_t<T, VEC>(GM + (size_t)r * a.d + c0, om);
      store_t<T, VEC>(GX + (size_t)r * a.d + c0, ox);
    }
    if (cl == 0 && r < a.rows) gs += gd * nx / nm;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gs += __shfl_xor(gs, off);
  if (lane == 0) red[wave] = gs;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) t += red[w];
    a.ws[blockIdx.x] = t;
  }
}

// one row per wave at most: fp32 d <= 256 (4 channels per lane), bf16 d <= 512 (8 channels per lane)
static bool mn_ok(int64_t d, int dtype) {
  if (dtype == MLGNN_DTYPE_F32) return d > 0 && d <= 256 && d % 4 == 0;
  if (dtype == MLGNN_DTYPE_BF16) return d > 0 && d <= 512 && d % 8 == 0;
  return false;
}

}  // namespace mlgnn

extern "C" int64_t mlgnn_msgnorm_bwd_workspace_floats(int64_t rows, int64_t d) {
  if (rows < 0 || d <= 0 || d > 512 || d % 4 != 0) return MLGNN_E_SHAPE;
  return mlgnn::ln_grid(rows, mlgnn::lanes_per_row_log2(d, 4));          // the fp32 layout's grid: the larger of the two
}

extern "C" int mlgnn_msgnorm_add_fwd(const void* x, const void* m, const float* scale, void* h,
                                     int64_t rows, int64_t d, int dtype, void* stream) {
  using namespace mlgnn;
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (rows < 0 || rows > INT32_MAX || !mn_ok(d, dtype)) return MLGNN_E_SHAPE;
  if (rows == 0) return 0;
  if (!x || !m || !scale || !h) return MLGNN_E_NULL;
  if (!a16(x) || !a16(m) || !a16(h)) return MLGNN_E_ALIGN;
  MnArgs a{};
  a.x = x; a.m = m; a.scale = scale; a.h = h; a.rows = (int)rows; a.d = (int)d;
  const int vec = dtype == MLGNN_DTYPE_BF16 ? 8 : 4;
  const int lpr = lanes_per_row_log2(d, vec);
  const dim3 grid(ln_grid(rows, lpr)), block(kBlock);
  if (dtype == MLGNN_DTYPE_BF16) {
    MLGNN_LNT_LAUNCH(msgnorm_add_fwd_kernel, bf16_t, 8, lpr, grid, block, 0, (hipStream_t)stream, a)
  } else {
    MLGNN_LNT_LAUNCH(msgnorm_add_fwd_kernel, float, 4, lpr, grid, block, 0, (hipStream_t)stream, a)
  }
  return (int)hipGetLastError();
}

extern "C" int mlgnn_msgnorm_add_bwd(const void* grad_h, const void* x, const void* m, const float* scale,
                                     void* grad_x, void* grad_m, float* grad_scale, float* workspace,
                                     int64_t workspace_floats, int64_t rows, int64_t d, int dtype, void* stream) {
  using namespace mlgnn;
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (rows < 0 || rows > INT32_MAX || !mn_ok(d, dtype)) return MLGNN_E_SHAPE;
  if (!grad_scale || !workspace || !scale) return MLGNN_E_NULL;
  const int vec = dtype == MLGNN_DTYPE_BF16 ? 8 : 4;
  const int lpr = lanes_per_row_log2(d, vec);
  const int nblk = ln_grid(rows, lpr);
  if (workspace_floats < nblk) return MLGNN_E_WORKSPACE;
  if (rows > 0 && (!grad_h || !x || !m || !grad_x || !grad_m)) return MLGNN_E_NULL;
  if (!a16(x) || !a16(m) || !a16(grad_h) || !a16(grad_x) || !a16(grad_m)) return MLGNN_E_ALIGN;
  MnArgs a{};
  a.x = x; a.m = m; a.gh = grad_h; a.scale = scale;
  a.gx = grad_x; a.gm = grad_m; a.ws = workspace; a.rows = (int)rows; a.d = (int)d;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MLGNN_DTYPE_BF16) {
    MLGNN_LNT_LAUNCH(msgnorm_add_bwd_kernel, bf16_t, 8, lpr, dim3(nblk), dim3(kBlock), 0, s, a)
  } else {
    MLGNN_LNT_LAUNCH(msgnorm_add_bwd_kernel, float, 4, lpr, dim3(nblk), dim3(kBlock), 0, s, a)
  }
  int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_scale, nblk, 1, s);
  return (int)hipGetLastError();
}
