// Streaming pieces of the SAGE update  out = leaky_relu([x | agg] W^T + b) * mask  (reference:
// models/gcn_lib/sparse/torch_vertex.py:288-291 `update`, torch_nn.py:9-24 `act_layer('leakyrelu', 0.2)`,
// multilevel_gnn.py:205-207 value_att_mask) that are not a GEMM epilogue:
//
//   leaky_relu_bwd:   dz = dy * mask[row] * (z > 0 ? 1 : slope)   with the sign of z recovered from the stored result
//                     y = leaky_relu(z) * mask (slope > 0: sign(y) = sign(z) sign(mask)); also max |dz| per row, the
//                     operand scale of the two input-gradient GEMMs and the weight gradient that consume dz
//   node_embed_fwd:   h[b, n, :] = x[b, n] * E[n, :]   (multilevel_gnn.py:151: the per-gene embedding scaled by the
//                     sample's value) + max |h| per row
//   node_embed_bwd:   dE[n, :] = sum_b x[b, n] * dh[b, n, :]   (one thread group per gene row, samples in order:
//                     bitwise reproducible, no atomics)
//
// All three are HBM-bound streams: 16-byte accesses, one pass, nothing kept.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using f4 = __attribute__((ext_vector_type(4))) float;
// read-once stream: non-temporal 16-byte load
__device__ __forceinline__ float4 stream_load4(const float4* p) {
  const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// LPR = lanes per row (J / 4): a power of two <= 64, so a row's lanes sit in one wave and its maximum is an xor butterfly
template <int LPR>
__global__ __launch_bounds__(256) void leaky_relu_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                             const float* __restrict__ row_scale, float slope,
                                                             float4* __restrict__ dz, float* __restrict__ row_max,
                                                             int64_t n_units) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // (n_units is a multiple of LPR and blockDim of 64: whole rows per wave, all lanes of a row in or out together)
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += stride) {
    const int64_t row = u / LPR;
    const float4 g = stream_load4(dy + u);
    const float4 v = y[u];
    const float sc = row_scale ? row_scale[row] : 1.f;
    const bool flip = sc < 0.f;
    float4 o;
    // z > 0  <=>  y and the mask have the same (non-zero) sign; a zero mask kills the gradient itself
    o.x = g.x * sc * ((flip ? v.x < 0.f : v.x > 0.f) ? 1.f : slope);
    o.y = g.y * sc * ((flip ? v.y < 0.f : v.y > 0.f) ? 1.f : slope);
    o.z = g.z * sc * ((flip ? v.z < 0.f : v.z > 0.f) ? 1.f : slope);
    o.w = g.w * sc * ((flip ? v.w < 0.f : v.w > 0.f) ? 1.f : slope);
    dz[u] = o;
    if (row_max) {
      float m = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
      if ((u & (LPR - 1)) == 0) row_max[row] = m;
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void node_embed_fwd_kernel(const float* __restrict__ x, const float4* __restrict__ emb,
                                                             float4* __restrict__ h, float* __restrict__ row_max,
                                                             int64_t n_units, int nodes) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += stride) {
    const int64_t row = u / LPR;
    const int n = (int)(row % nodes);
    const float s = x[row];
    const float4 e = emb[(int64_t)n * LPR + (u & (LPR - 1))];
    const float4 o = make_float4(e.x * s, e.y * s, e.z * s, e.w * s);
    h[u] = o;
    if (row_max) {
      float m = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
      if ((u & (LPR - 1)) == 0) row_max[row] = m;
    }
  }
}

// one thread per (gene, column quad): its B cotangent rows are `nodes` rows apart -- independent 16-byte loads, 8 in
// flight per thread
template <int LPR>
__global__ __launch_bounds__(256) void node_embed_bwd_kernel(const float* __restrict__ x, const float4* __restrict__ dh,
                                                             float4* __restrict__ demb, int nodes, int batch) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (int64_t)nodes * LPR) return;
  const int n = (int)(u / LPR);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t step = (int64_t)nodes * LPR;
  int b = 0;
  for (; b + 8 <= batch; b += 8) {
    float4 g[8];
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      g[i] = stream_load4(dh + u + (int64_t)(b + i) * step);
      s[i] = x[(int64_t)(b + i) * nodes + n];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc.x = fmaf(g[i].x, s[i], acc.x); acc.y = fmaf(g[i].y, s[i], acc.y);
      acc.z = fmaf(g[i].z, s[i], acc.z); acc.w = fmaf(g[i].w, s[i], acc.w);
    }
  }
  for (; b < batch; ++b) {
    const float4 g = dh[u + (int64_t)b * step];
    const float s = x[(int64_t)b * nodes + n];
    acc.x = fmaf(g.x, s, acc.x); acc.y = fmaf(g.y, s, acc.y); acc.z = fmaf(g.z, s, acc.z); acc.w = fmaf(g.w, s, acc.w);
  }
  demb[u] = acc;
}

// Streaming ceiling of the box (bench.py `roofline.stream_copy_GBps`): 16 bytes per lane, non-temporal both ways, eight
// independent loads in flight per thread -- the shape MI355X_MICROARCH.md quotes its float4-copy figure for.
template <bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n) {
  // a workgroup moves contiguous 32 KB pieces (8 x 4 KB, one 16-byte access per lane each), pieces dealt round robin
  const int64_t pieces = n / 2048;
  for (int64_t p = blockIdx.x; p < pieces; p += gridDim.x) {
    const int64_t base = p * 2048 + threadIdx.x;
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = NT ? stream_load4(src + base + q * 256) : src[base + q * 256];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if constexpr (NT) {
        const f4 t = {v[q].x, v[q].y, v[q].z, v[q].w};
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(dst + base + q * 256));
      } else {
        dst[base + q * 256] = v[q];
      }
    }
  }
  for (int64_t i = pieces * 2048 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

// ---- Linear with a handful of input columns (the node encoder Linear(3, hidden) of deepergcn.py:199-210: x [N, 3]) ----
// forward: one 16-byte store per thread, the R <= 8 inputs of the row broadcast to its lanes, W[4 columns][R] in registers.
// A write-bound stream (328 MB at BASELINE configs[1]) that the library ran as a K = 3 GEMM (100 us).
template <int LPR, int R>
__global__ __launch_bounds__(256) void narrow_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float4* __restrict__ out,
                                                                int64_t n_units) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int q = threadIdx.x & (LPR - 1);                 // (blockDim and the grid stride are multiples of LPR)
  float wr[4][R], b4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    b4[c] = bias ? bias[4 * q + c] : 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) wr[c][k] = w[(4 * q + c) * R + k];
  }
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += stride) {
    const int64_t row = u / LPR;
    float xv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) xv[k] = x[row * R + k];
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o[c] = b4[c];
#pragma unroll
      for (int k = 0; k < R; ++k) o[c] = fmaf(xv[k], wr[c][k], o[c]);
    }
    const f4 t = {o[0], o[1], o[2], o[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f4*>(out + u));
  }
}

// weight + bias gradient: a thread keeps its 4 columns x (R + 1) sums over the rows it walks; the row groups of a
// workgroup are folded through LDS in order, one [J (R + 1)] partial per workgroup, reduced in a fixed order afterwards
template <int LPR, int R>
__global__ __launch_bounds__(256) void narrow_linear_bwd_kernel(const float4* __restrict__ go, const float* __restrict__ x,
                                                                float* __restrict__ partial, int64_t n_units) {
  __shared__ float red[256 * 4 * (R + 1)];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int q = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  constexpr int kGroups = 256 / LPR, J = 4 * LPR;
  float acc[4][R + 1];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k <= R; ++k) acc[c][k] = 0.f;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += stride) {
    const int64_t row = u / LPR;
    const float4 g = stream_load4(go + u);
    const float gv[4] = {g.x, g.y, g.z, g.w};
    float xv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) xv[k] = x[row * R + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int k = 0; k < R; ++k) acc[c][k] = fmaf(gv[c], xv[k], acc[c][k]);
      acc[c][R] += gv[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k <= R; ++k) red[(grp * J + 4 * q + c) * (R + 1) + k] = acc[c][k];
  __syncthreads();
  for (int i = threadIdx.x; i < J * (R + 1); i += 256) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < kGroups; ++g2) t += red[g2 * J * (R + 1) + i];
    const int j = i / (R + 1), k = i - j * (R + 1);
    // layout of the result: grad_w [J, R] then grad_b [J]
    partial[(size_t)blockIdx.x * (J * (R + 1)) + (k < R ? j * R + k : J * R + j)] = t;
  }
}

constexpr int kNarrowBlocks = 1024;

// ---- the fold of lin_r into the SAGE update's weight (mlgnn/sage.py) and its chain rule, on [out, in]-sized matrices ----
// forward:  W_c = W_a W_r  (W_nn = [W_x | W_a], W_a [out, out], W_r [out, in]);  w_cat = [W_x - rel W_c | W_c]  [out, 2 in],
//           plus the two halves as matrices of their own (the operands of the backward's input-gradient GEMMs)
__global__ __launch_bounds__(256) void sage_fold_fwd_kernel(const float* __restrict__ w_nn, const float* __restrict__ w_r,
                                                           float* __restrict__ w_cat, float* __restrict__ w_x1,
                                                           float* __restrict__ w_c, int cin, int cout, int relative) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= cout * cin) return;
  const int o = idx / cin, i = idx - o * cin;
  const float* wa = w_nn + (size_t)o * (cin + cout) + cin;
  float c = 0.f;
  for (int k = 0; k < cout; ++k) c = fmaf(wa[k], w_r[(size_t)k * cin + i], c);
  const float x1 = w_nn[(size_t)o * (cin + cout) + i] - (relative ? c : 0.f);
  w_cat[(size_t)o * 2 * cin + i] = x1;
  w_cat[(size_t)o * 2 * cin + cin + i] = c;
  w_x1[idx] = x1;
  w_c[idx] = c;
}

// backward:  G_c = gw_c - rel gw_x1;  g_nn = [gw_x1 | G_c W_r^T]  [out, in + out];  g_r = W_a^T G_c  [out, in]
__global__ __launch_bounds__(256) void sage_fold_bwd_kernel(const float* __restrict__ gw_x1, const float* __restrict__ gw_c,
                                                           const float* __restrict__ w_nn, const float* __restrict__ w_r,
                                                           float* __restrict__ g_nn, float* __restrict__ g_r, int cin, int cout,
                                                           int relative) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int n_nn = cout * (cin + cout), n_r = cout * cin;
  if (idx < n_nn) {
    const int o = idx / (cin + cout), j = idx - o * (cin + cout);
    if (j < cin) {
      g_nn[idx] = gw_x1[(size_t)o * cin + j];
    } else {
      const int k = j - cin;
      float t = 0.f;
      for (int i = 0; i < cin; ++i) {
        const float g = gw_c[(size_t)o * cin + i] - (relative ? gw_x1[(size_t)o * cin + i] : 0.f);
        t = fmaf(g, w_r[(size_t)k * cin + i], t);
      }
      g_nn[idx] = t;
    }
  } else if (idx < n_nn + n_r) {
    const int e = idx - n_nn, k = e / cin, i = e - k * cin;
    float t = 0.f;
    for (int o = 0; o < cout; ++o) {
      const float g = gw_c[(size_t)o * cin + i] - (relative ? gw_x1[(size_t)o * cin + i] : 0.f);
      t = fmaf(w_nn[(size_t)o * (cin + cout) + cin + k], g, t);
    }
    g_r[e] = t;
  }
}


// ---- batched transpose [B, R, C] -> [B, C, R] (fp32) -------------------------------------------------------------------
// The flatten in front of MultilevelGNN's first head Linear (multilevel_gnn.py:277: torch.flatten of the [B, C, 146, 3k]
// convolution result) when the result lives channel-last (the layout the 1x1 convolutions compute in): a 64 x 64 tile through
// LDS, both sides in 256-byte pieces.  The backward is the same kernel the other way round.
constexpr int kTrTile = 64;
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
  __shared__ float tile[kTrTile][kTrTile + 1];
  const int b = blockIdx.z, r0 = blockIdx.x * kTrTile, c0 = blockIdx.y * kTrTile;
  const float* s = src + (size_t)b * R * C;
  float* d = dst + (size_t)b * R * C;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;             // 64 x 4
  for (int i = ty; i < kTrTile; i += 4)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = s[(size_t)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < kTrTile; i += 4)
    if (c0 + i < C && r0 + tx < R) d[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

static bool width_ok(int64_t J) {
  const int64_t l = J / 4;
  return J >= 4 && J % 4 == 0 && l <= 64 && (l & (l - 1)) == 0;
}

static int stream_grid(int64_t n_units) {
  int64_t b = (n_units + 255) / 256;
  const int64_t cap = 256 * 16;                       // 16 workgroups of 4 waves per CU: enough loads in flight, few tails
  return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace mlgnn

using namespace mlgnn;

#define MLGNN_LPR_SWITCH(LPR_, BODY)                  \
  switch (LPR_) {                                     \
    case 1: { constexpr int L = 1; BODY } break;      \
    case 2: { constexpr int L = 2; BODY } break;      \
    case 4: { constexpr int L = 4; BODY } break;      \
    case 8: { constexpr int L = 8; BODY } break;      \
    case 16: { constexpr int L = 16; BODY } break;    \
    case 32: { constexpr int L = 32; BODY } break;    \
    default: { constexpr int L = 64; BODY } break;    \
  }

extern "C" int mlgnn_leaky_relu_bwd(const float* grad_out, const float* y, const float* row_scale, float slope,
                                    float* grad_z, float* grad_z_row_max, int64_t N, int64_t J, void* stream) {
  if (N < 0 || N > INT32_MAX || !width_ok(J)) return MLGNN_E_SHAPE;
  if (N == 0) return 0;
  if (!grad_out || !y || !grad_z) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(grad_z)) & 15) != 0)
    return MLGNN_E_ALIGN;
  const int64_t units = N * (J / 4);
  hipStream_t s = (hipStream_t)stream;
  MLGNN_LPR_SWITCH((int)(J / 4), hipLaunchKernelGGL((leaky_relu_bwd_kernel<L>), dim3(stream_grid(units)), dim3(256), 0, s,
                   reinterpret_cast<const float4*>(grad_out), reinterpret_cast<const float4*>(y), row_scale, slope,
                   reinterpret_cast<float4*>(grad_z), grad_z_row_max, units);)
  return (int)hipGetLastError();
}

extern "C" int mlgnn_node_embed_fwd(const float* x, const float* embedding, float* h, float* h_row_max, int64_t batch,
                                    int64_t nodes, int64_t C, void* stream) {
  if (batch < 0 || nodes <= 0 || nodes > INT32_MAX || batch * nodes > INT32_MAX || !width_ok(C)) return MLGNN_E_SHAPE;
  if (batch == 0) return 0;
  if (!x || !embedding || !h) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(embedding) | reinterpret_cast<uintptr_t>(h)) & 15) != 0) return MLGNN_E_ALIGN;
  const int64_t units = batch * nodes * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  MLGNN_LPR_SWITCH((int)(C / 4), hipLaunchKernelGGL((node_embed_fwd_kernel<L>), dim3(stream_grid(units)), dim3(256), 0, s, x,
                   reinterpret_cast<const float4*>(embedding), reinterpret_cast<float4*>(h), h_row_max, units, (int)nodes);)
  return (int)hipGetLastError();
}

extern "C" int mlgnn_node_embed_bwd(const float* x, const float* grad_h, float* grad_embedding, int64_t batch,
                                    int64_t nodes, int64_t C, void* stream) {
  if (batch < 0 || batch > INT32_MAX || nodes <= 0 || nodes > INT32_MAX || batch * nodes > INT32_MAX || !width_ok(C))
    return MLGNN_E_SHAPE;
  if (!grad_embedding) return MLGNN_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (batch == 0) return (int)hipMemsetAsync(grad_embedding, 0, (size_t)(nodes * C) * sizeof(float), s);
  if (!x || !grad_h) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_h) | reinterpret_cast<uintptr_t>(grad_embedding)) & 15) != 0) return MLGNN_E_ALIGN;
  const int64_t units = nodes * (C / 4);
  MLGNN_LPR_SWITCH((int)(C / 4), hipLaunchKernelGGL((node_embed_bwd_kernel<L>), dim3((unsigned)((units + 255) / 256)), dim3(256),
                   0, s, x, reinterpret_cast<const float4*>(grad_h), reinterpret_cast<float4*>(grad_embedding), (int)nodes,
                   (int)batch);)
  return (int)hipGetLastError();
}

extern "C" int mlgnn_stream_copy(const void* src, void* dst, int64_t bytes, int non_temporal, void* stream) {
  if (bytes < 0 || bytes % 16 != 0) return MLGNN_E_SHAPE;
  if (bytes == 0) return 0;
  if (!src || !dst) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0) return MLGNN_E_ALIGN;
  if (non_temporal)
    hipLaunchKernelGGL(stream_copy_kernel<true>, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), bytes / 16);
  else
    hipLaunchKernelGGL(stream_copy_kernel<false>, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), bytes / 16);
  return (int)hipGetLastError();
}

#define MLGNN_R_SWITCH(R_, BODY)                    \
  switch (R_) {                                     \
    case 1: { constexpr int RR = 1; BODY } break;   \
    case 2: { constexpr int RR = 2; BODY } break;   \
    case 3: { constexpr int RR = 3; BODY } break;   \
    case 4: { constexpr int RR = 4; BODY } break;   \
    case 5: { constexpr int RR = 5; BODY } break;   \
    case 6: { constexpr int RR = 6; BODY } break;   \
    case 7: { constexpr int RR = 7; BODY } break;   \
    default: { constexpr int RR = 8; BODY } break;  \
  }

static bool narrow_ok(int64_t N, int64_t R, int64_t J) {
  // J = 4 * lanes with 8, 16, 32 or 64 lanes per row (32 .. 256 columns); 1 .. 8 input columns
  return N >= 0 && N <= INT32_MAX && R >= 1 && R <= 8 && (J == 32 || J == 64 || J == 128 || J == 256);
}

extern "C" int mlgnn_narrow_linear_supported(int64_t N, int64_t R, int64_t J) { return narrow_ok(N, R, J) ? 1 : 0; }

extern "C" int64_t mlgnn_narrow_linear_bwd_workspace_floats(int64_t R, int64_t J) {
  if (!narrow_ok(1, R, J)) return MLGNN_E_SHAPE;
  return (int64_t)kNarrowBlocks * J * (R + 1);
}

#define MLGNN_NARROW_LPR(J_, BODY)                      \
  switch ((int)(J_ / 4)) {                              \
    case 8: { constexpr int L = 8; BODY } break;        \
    case 16: { constexpr int L = 16; BODY } break;      \
    case 32: { constexpr int L = 32; BODY } break;      \
    default: { constexpr int L = 64; BODY } break;      \
  }

extern "C" int mlgnn_narrow_linear_fwd(const float* x, const float* w, const float* bias, float* out, int64_t N, int64_t R,
                                       int64_t J, void* stream) {
  if (!narrow_ok(N, R, J)) return MLGNN_E_SHAPE;
  if (N == 0) return 0;
  if (!x || !w || !out) return MLGNN_E_NULL;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return MLGNN_E_ALIGN;
  const int64_t units = N * (J / 4);
  hipStream_t s = (hipStream_t)stream;
  MLGNN_NARROW_LPR(J, MLGNN_R_SWITCH((int)R, hipLaunchKernelGGL((narrow_linear_fwd_kernel<L, RR>), dim3(stream_grid(units)),
                   dim3(256), 0, s, x, w, bias, reinterpret_cast<float4*>(out), units);))
  return (int)hipGetLastError();
}

extern "C" int mlgnn_narrow_linear_bwd(const float* grad_out, const float* x, float* grad_w_b, float* workspace,
                                       int64_t workspace_floats, int64_t N, int64_t R, int64_t J, void* stream) {
  if (!narrow_ok(N, R, J)) return MLGNN_E_SHAPE;
  if (!grad_w_b || !workspace) return MLGNN_E_NULL;
  if (N > 0 && (!grad_out || !x)) return MLGNN_E_NULL;
  if (workspace_floats < mlgnn_narrow_linear_bwd_workspace_floats(R, J)) return MLGNN_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(grad_out) & 15) != 0) return MLGNN_E_ALIGN;
  const int64_t units = N * (J / 4);
  hipStream_t s = (hipStream_t)stream;
  MLGNN_NARROW_LPR(J, MLGNN_R_SWITCH((int)R, hipLaunchKernelGGL((narrow_linear_bwd_kernel<L, RR>), dim3(kNarrowBlocks), dim3(256),
                   0, s, reinterpret_cast<const float4*>(grad_out), x, workspace, units);))
  const int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_w_b, kNarrowBlocks, (int)(J * (R + 1)), s);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_sage_fold_fwd(const float* w_nn, const float* w_r, float* w_cat, float* w_x1, float* w_c, int64_t cin,
                                   int64_t cout, int relative, void* stream) {
  if (cin < 1 || cout < 1 || cin > 4096 || cout > 4096) return MLGNN_E_SHAPE;
  if (!w_nn || !w_r || !w_cat || !w_x1 || !w_c) return MLGNN_E_NULL;
  hipLaunchKernelGGL(sage_fold_fwd_kernel, dim3((unsigned)((cin * cout + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_nn,
                     w_r, w_cat, w_x1, w_c, (int)cin, (int)cout, relative);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_sage_fold_bwd(const float* grad_w_x1, const float* grad_w_c, const float* w_nn, const float* w_r,
                                   float* grad_w_nn, float* grad_w_r, int64_t cin, int64_t cout, int relative, void* stream) {
  if (cin < 1 || cout < 1 || cin > 4096 || cout > 4096) return MLGNN_E_SHAPE;
  if (!grad_w_x1 || !grad_w_c || !w_nn || !w_r || !grad_w_nn || !grad_w_r) return MLGNN_E_NULL;
  const int64_t n = cout * (cin + cout) + cout * cin;
  hipLaunchKernelGGL(sage_fold_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_w_x1,
                     grad_w_c, w_nn, w_r, grad_w_nn, grad_w_r, (int)cin, (int)cout, relative);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_transpose_batched(const float* src, float* dst, int64_t B, int64_t R, int64_t C, void* stream) {
  if (B < 0 || R < 0 || C < 0 || B > 65535 || R > INT32_MAX || C > INT32_MAX || (C + kTrTile - 1) / kTrTile > 65535)
    return MLGNN_E_SHAPE;
  if (B == 0 || R == 0 || C == 0) return 0;
  if (!src || !dst) return MLGNN_E_NULL;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)((R + kTrTile - 1) / kTrTile), (unsigned)((C + kTrTile - 1) / kTrTile), (unsigned)B),
                     dim3(256), 0, (hipStream_t)stream, src, dst, (int)R, (int)C);
  return (int)hipGetLastError();
}
