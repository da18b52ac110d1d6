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

// ---- table gradient through a fixed-point accumulator (max aggregation, csrc/aggregate_bwd.hip) ---------------------
// begin: header = {bits of max |go| over the cotangent the aggregation backward is about to read, non-finite flag,
// headroom bits}; the aggregation backward then adds round(dz * scale) with integer atomics; finish: the sums leave as
// fp32 (total = or += sum / scale) and the accumulator is cleared for the next layer.
__global__ __launch_bounds__(256) void fix_absmax_kernel(const float4* __restrict__ go, int64_t n4, uint32_t* __restrict__ hdr,
                                                         int bits) {
  __shared__ float wmax[4];
  __shared__ int wbad[4];
  float m = 0.f;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = go[i];
    const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (a[k] <= 3.4028234e38f) m = fmaxf(m, a[k]);
      else bad = 1;                                              // Inf or NaN
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o));
    bad |= __shfl_xor(bad, o);
  }
  if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = m; wbad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    bad = wbad[0] | wbad[1] | wbad[2] | wbad[3];
    atomicMax(&hdr[0], __builtin_bit_cast(uint32_t, m));         // non-negative floats order like their bit patterns
    if (bad) atomicOr(&hdr[1], 1u);
    if (blockIdx.x == 0) hdr[2] = (uint32_t)bits;
  }
}

__global__ __launch_bounds__(256) void fix_to_table_kernel(long long* __restrict__ fix, const uint32_t* __restrict__ hdr,
                                                           float* __restrict__ total, int64_t n, int accumulate) {
  const float inv = __builtin_bit_cast(float, (uint32_t)(254 - fix_scale_exponent(hdr)) << 23);      // 1 / scale, exact
  const bool bad = hdr[1] != 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long q = fix[i];
    float v = bad ? __builtin_nanf("") : __ll2float_rn(q) * inv;
    total[i] = accumulate ? total[i] + v : v;
    fix[i] = 0;
  }
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_table_grad_bytes(int64_t T, int64_t d) {
  if (T <= 0 || d <= 0 || d % 4 != 0 || d > 4096) return MLGNN_E_SHAPE;
  return (int64_t)kFixHeaderBytes + T * d * 8;
}

extern "C" int mlgnn_table_grad_begin(const float* grad_out, int64_t rows, int64_t d, void* accumulator, void* stream) {
  if (rows <= 0 || d <= 0 || d % 4 != 0 || rows * d > ((int64_t)1 << 40)) return MLGNN_E_SHAPE;
  if (!grad_out || !accumulator) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(accumulator)) & 15) != 0) return MLGNN_E_ALIGN;
  // at most `rows` winners add to one table entry: |sum| < rows * 2^bits must stay below 2^62
  int lg = 0;
  while (((int64_t)1 << lg) < rows + 1) ++lg;
  const int bits = 62 - lg;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(accumulator, 0, 16, s);
  if (e != hipSuccess) return (int)e;
  const int64_t n4 = rows * d / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fix_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(grad_out), n4,
                     static_cast<uint32_t*>(accumulator), bits);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_table_grad_finish(void* accumulator, float* grad_table, int64_t T, int64_t d, int accumulate,
                                       void* stream) {
  if (T <= 0 || d <= 0 || d % 4 != 0) return MLGNN_E_SHAPE;
  if (!accumulator || !grad_table) return MLGNN_E_NULL;
  const int64_t n = T * d;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  unsigned char* base = static_cast<unsigned char*>(accumulator);
  hipLaunchKernelGGL(fix_to_table_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<long long*>(base + kFixHeaderBytes), reinterpret_cast<const uint32_t*>(base), grad_table, n,
                     accumulate);
  return (int)hipGetLastError();
}

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
