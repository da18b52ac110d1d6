// One 16x16 output tile on v_mfma_f32_16x16x4_f32 with operands fetched through accessors, for the
// small pooled-graph kernels (diffpool.hip, densesage.hip): lane l supplies A[i = l & 15][k = l >> 4]
// and B[k = l >> 4][j = l & 15] per k-step of 4; C/D: col = l & 15, row = 4 * (l >> 4) + reg.
#pragma once
#include "common.h"

namespace mlgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// acc[i][j] = sum_k a_at(i, k) * b_at(k, j).  The accessors return 0 for every k at or past the end of
// their operand (the loop runs in blocks of kTileUnroll k-steps: all operand loads of a block are issued
// before its MFMAs, so their latencies overlap instead of adding up).
constexpr int kTileUnroll = 8;

template <typename FA, typename FB>
__device__ __forceinline__ f32x4 tile_gemm(int kdim, FA a_at, FB b_at) {
  const int lane = threadIdx.x & (kWave - 1);
  const int l15 = lane & 15, lk = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kdim; k0 += 4 * kTileUnroll) {
    float a[kTileUnroll], b[kTileUnroll];
#pragma unroll
    for (int u = 0; u < kTileUnroll; ++u) {
      a[u] = a_at(l15, k0 + 4 * u + lk);
      b[u] = b_at(k0 + 4 * u + lk, l15);
    }
#pragma unroll
    for (int u = 0; u < kTileUnroll; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
  }
  return acc;
}

}  // namespace mlgnn
