// Weighted sum / mean aggregation for NARROW features over SHORT rows: one lane group per node row.
//
// The GraphSAGE layers of the shipped configs (config/kirc.yaml, gbm.yaml: gnn_name sage; reference
// models/gcn_lib/sparse/torch_vertex.py:269-294) aggregate 32- and 64-channel rows over the TCGA gene network -- 60 000
// edges on 15 405 nodes, 4.9 edges per row with the added self loop.  The general kernels (aggregate_fwd.hip /
// aggregate_bwd.hip) give a whole wavefront to one row and split its edges over the wave's lane groups: at 5 edges and
// 8 lanes per 128-byte row, most of the wave idles through a chain of dependent loads (row pointer -> column -> row) per
// row: 278 / 317 us per launch at kirc shape.  Here a row belongs to ONE group of LPR = d / 4 lanes (1..16: d <= 64
// channels) and a wave walks 64 / LPR rows at once, each group keeping four gathered rows in flight; the rows of a
// workgroup are consecutive (one contiguous block of stores) and the XCDs own contiguous eighths of the row range, like
// the chunk walk of the general kernels.  Same contract as their main launch -- rows are clamped to their first `cap`
// edges (long rows: the chunk launches and the combine of csrc/hub.hip follow unchanged) -- so which kernel runs depends
// on the shape only, never on whether the graph's long-row tables are known yet.  No atomics: bitwise reproducible.
#pragma once
#include "aggregate_common.h"

namespace mlgnn {

constexpr int kShortBlock = 256;

// rows [first, last) of this workgroup: XCD x owns the x-th eighth of the rows, workgroup slot b / 8 a block of them
__device__ __forceinline__ void short_rows_of_block(int n_rows, int rows_per_block, int& first, int& last) {
  const int xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds;
  int per_xcd = (n_rows + kXcds - 1) / kXcds;
  per_xcd = (per_xcd + rows_per_block - 1) / rows_per_block * rows_per_block;
  const int lo = xcd * per_xcd, hi = min(n_rows, lo + per_xcd);
  first = lo + slot * rows_per_block;
  last = min(hi, first + rows_per_block);
}

inline int short_grid(int64_t n_rows, int rows_per_block) {
  int64_t per_xcd = (n_rows + kXcds - 1) / kXcds;
  per_xcd = (per_xcd + rows_per_block - 1) / rows_per_block;       // blocks per XCD
  return (int)(per_xcd * kXcds);
}

// forward: out[r] = (mean ? 1 / deg : 1) * sum_{e in row r} (w_e) x[col_e]
template <int LPR, bool WEIGHTED>
__global__ __launch_bounds__(kShortBlock) void csr_short_fwd_kernel(const float* __restrict__ x, const int* __restrict__ rowptr,
                                                                   const int* __restrict__ col, const float* __restrict__ ew,
                                                                   float* __restrict__ out, int N, int mean, int cap) {
  constexpr int kRows = kShortBlock / LPR;
  constexpr int D = 4 * LPR;
  int first, last;
  short_rows_of_block(N, kRows, first, last);
  const int r = first + threadIdx.x / LPR, cl = threadIdx.x % LPR;
  if (r >= last) return;
  const int beg = rowptr[r];
  const int end = min(rowptr[r + 1], beg + cap);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* xc = x + 4 * cl;
  int e = beg;
  for (; e + 4 <= end; e += 4) {
    int c[4];
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { c[u] = col[e + u]; w[u] = WEIGHTED ? ew[e + u] : 1.f; }
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xc + (size_t)c[u] * D);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = fmaf(v[u].x, w[u], acc[0]); acc[1] = fmaf(v[u].y, w[u], acc[1]);
      acc[2] = fmaf(v[u].z, w[u], acc[2]); acc[3] = fmaf(v[u].w, w[u], acc[3]);
    }
  }
  for (; e < end; ++e) {
    const float w = WEIGHTED ? ew[e] : 1.f;
    const float4 v = *reinterpret_cast<const float4*>(xc + (size_t)col[e] * D);
    acc[0] = fmaf(v.x, w, acc[0]); acc[1] = fmaf(v.y, w, acc[1]); acc[2] = fmaf(v.z, w, acc[2]); acc[3] = fmaf(v.w, w, acc[3]);
  }
  const int deg = end - beg;
  const float s = (mean && deg > 0) ? 1.f / (float)deg : 1.f;
  *reinterpret_cast<float4*>(out + (size_t)r * D + 4 * cl) = make_float4(acc[0] * s, acc[1] * s, acc[2] * s, acc[3] * s);
}

// backward (by-source CSR): gx[j] = sum_{e: src = j} w_e * (mean ? 1 / indeg(dst_e) : 1) * go[dst_e]
template <int LPR, bool WEIGHTED>
__global__ __launch_bounds__(kShortBlock) void csr_short_bwd_kernel(const float* __restrict__ go, const int* __restrict__ rowptr_t,
                                                                   const int* __restrict__ col_t, const float* __restrict__ ew_t,
                                                                   const int* __restrict__ rowptr, float* __restrict__ gx, int N,
                                                                   int mean, int cap) {
  constexpr int kRows = kShortBlock / LPR;
  constexpr int D = 4 * LPR;
  int first, last;
  short_rows_of_block(N, kRows, first, last);
  const int r = first + threadIdx.x / LPR, cl = threadIdx.x % LPR;
  if (r >= last) return;
  const int beg = rowptr_t[r];
  const int end = min(rowptr_t[r + 1], beg + cap);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* gc = go + 4 * cl;
  auto scale_of = [&](int dst, float w) {
    if (!mean) return w;
    const int dg = rowptr[dst + 1] - rowptr[dst];       // (>= 1: this very edge ends there)
    return w / (float)dg;
  };
  int e = beg;
  for (; e + 4 <= end; e += 4) {
    int c[4];
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { c[u] = col_t[e + u]; w[u] = WEIGHTED ? ew_t[e + u] : 1.f; }
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(gc + (size_t)c[u] * D);
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = scale_of(c[u], w[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = fmaf(v[u].x, w[u], acc[0]); acc[1] = fmaf(v[u].y, w[u], acc[1]);
      acc[2] = fmaf(v[u].z, w[u], acc[2]); acc[3] = fmaf(v[u].w, w[u], acc[3]);
    }
  }
  for (; e < end; ++e) {
    const int c = col_t[e];
    const float w = scale_of(c, WEIGHTED ? ew_t[e] : 1.f);
    const float4 v = *reinterpret_cast<const float4*>(gc + (size_t)c * D);
    acc[0] = fmaf(v.x, w, acc[0]); acc[1] = fmaf(v.y, w, acc[1]); acc[2] = fmaf(v.z, w, acc[2]); acc[3] = fmaf(v.w, w, acc[3]);
  }
  *reinterpret_cast<float4*>(gx + (size_t)r * D + 4 * cl) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// d in {4, 8, 16, 32, 64}: lane groups of 1 .. 16 lanes
inline bool short_width_ok(int64_t d) { return d == 4 || d == 8 || d == 16 || d == 32 || d == 64; }

#define MLGNN_SHORT_DISPATCH(KERNEL, d_, weighted_, ...)                                        \
  do {                                                                                          \
    const dim3 sg_(short_grid(N, kShortBlock / (int)((d_) / 4))), sb_(kShortBlock);             \
    switch ((int)((d_) / 4)) {                                                                  \
      case 1: if (weighted_) hipLaunchKernelGGL((KERNEL<1, true>), sg_, sb_, 0, s, __VA_ARGS__);  \
              else hipLaunchKernelGGL((KERNEL<1, false>), sg_, sb_, 0, s, __VA_ARGS__); break;    \
      case 2: if (weighted_) hipLaunchKernelGGL((KERNEL<2, true>), sg_, sb_, 0, s, __VA_ARGS__);  \
              else hipLaunchKernelGGL((KERNEL<2, false>), sg_, sb_, 0, s, __VA_ARGS__); break;    \
      case 4: if (weighted_) hipLaunchKernelGGL((KERNEL<4, true>), sg_, sb_, 0, s, __VA_ARGS__);  \
              else hipLaunchKernelGGL((KERNEL<4, false>), sg_, sb_, 0, s, __VA_ARGS__); break;    \
      case 8: if (weighted_) hipLaunchKernelGGL((KERNEL<8, true>), sg_, sb_, 0, s, __VA_ARGS__);  \
              else hipLaunchKernelGGL((KERNEL<8, false>), sg_, sb_, 0, s, __VA_ARGS__); break;    \
      default: if (weighted_) hipLaunchKernelGGL((KERNEL<16, true>), sg_, sb_, 0, s, __VA_ARGS__); \
               else hipLaunchKernelGGL((KERNEL<16, false>), sg_, sb_, 0, s, __VA_ARGS__); break;  \
    }                                                                                           \
  } while (0)

}  // namespace mlgnn
