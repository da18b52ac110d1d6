// Gradient of an embedding lookup  e = table[idx]  for a dense [E, d] cotangent:
//     grad_table[t, :] = sum over { e : idx[e] = t } of grad_e[e, :]
//
// Reference: the edge-type embedding of DeeperGCN (models/deepergcn.py:103-104,189-190,213: global_edge='onehot',
// nn.Embedding(pathway_edge_num, hidden) applied to every edge) -- the autograd of that lookup.  ATen sorts the indices
// and then runs a segmented reduction that streams the cotangent at ~2 TB/s; here the rows of one table entry are
// gathered by one wavefront through a (stable) sorted permutation and summed in a fixed order: deterministic, one
// pass over grad_e (E d 4 bytes) at gather speed.
//
// Layout: lane groups of d/4 lanes hold one cotangent row (float4 per lane), 64 / (d/4) rows per wave and load
// instruction, four load instructions in flight; 64-bit row offsets (the cotangent of a 10 M-edge batch is 5 GB).
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

struct EmbArgs {
  const float* ge; const int* perm; const int* rowptr; float* out;
  int T; int d; int lpr_log2;
};

__global__ __launch_bounds__(kBlock) void embedding_grad_kernel(const EmbArgs a) {
  constexpr int kUnroll = 4;
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << a.lpr_log2, groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2, cl = lane & (lpr - 1);
  const int wave_global = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  for (int cbase = 0; cbase < a.d; cbase += lpr * 4) {
    const int c0 = min(cbase + cl * 4, a.d - 4);
    const bool cact = cbase + cl * 4 < a.d;
    for (int t = wave_global; t < a.T; t += n_waves) {
      const int beg = a.rowptr[t], end = a.rowptr[t + 1];
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = beg + sub; k < end; k += groups * kUnroll) {
        float v[kUnroll][4];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int kk = k + u * groups;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[u][i] = 0.f;
          if (kk < end) load_vec<4>(v[u], a.ge + (size_t)a.perm[kk] * a.d + c0);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += v[u][i];
      }
      for (int off = lpr; off < kWave; off <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off);
      if (sub == 0 && cact) store_vec<4>(a.out + (size_t)t * a.d + c0, acc);
    }
  }
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_embedding_bwd(const float* grad_e, const int32_t* perm, const int32_t* rowptr, float* grad_table,
                                   int64_t T, int64_t d, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32) return MLGNN_E_DTYPE;
  if (T < 0 || T > INT32_MAX || d <= 0 || d % 4 != 0 || d > 4096) return MLGNN_E_SHAPE;
  if (T == 0) return 0;
  if (!grad_e || !perm || !rowptr || !grad_table) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_e) | reinterpret_cast<uintptr_t>(grad_table)) & 15) != 0) return MLGNN_E_ALIGN;
  EmbArgs a;
  a.ge = grad_e; a.perm = perm; a.rowptr = rowptr; a.out = grad_table; a.T = (int)T; a.d = (int)d;
  a.lpr_log2 = lanes_per_row_log2(d, 4);
  int64_t blocks = (T + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  hipLaunchKernelGGL(embedding_grad_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
