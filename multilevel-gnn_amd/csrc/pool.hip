// Per-graph readout over the contiguous node ranges of a batch: out[b,:] = reduce_{n in graph b} x[n,:].
//
// Reference: torch_geometric global_{add,mean,max}_pool called from models/deepergcn.py:148-155,319
// (scatter over the `batch` vector; an empty graph gives 0; max keeps the first maximal row like
// torch_scatter's CPU path).  Graph b owns rows [ptr[b], ptr[b+1]) (PyG batches are sorted by graph).
// Two deterministic stages, no atomics: (graph, slice) workgroups reduce a row slice each, then the
// slices of a graph are folded in order.  HBM-bound: reads N*d*4 bytes once.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

// row slices per graph: 8 for batches of many graphs, more when the batch has few (configs[4]: one graph of
// 200 000 nodes would otherwise be read by 8 workgroups), about 1024 workgroups in total
static int pool_slices(int64_t B) {
  int64_t s = 1024 / (B > 0 ? B : 1);
  return (int)(s < 8 ? 8 : (s > 256 ? 256 : s));
}
constexpr float kPoolNegBig = -3.0e38f;

struct PoolArgs {
  const float* x; const int* ptr; float* part; int* part_arg; float* out; int* argmax;
  int B; int d; int lpr_log2; int kind; int slices;      // kind: 0 sum, 1 mean, 2 max
};

__global__ __launch_bounds__(kBlock) void segment_pool_stage1_kernel(const PoolArgs a) {
  __shared__ float redv[kWavesPerBlock][kWave * 4];
  __shared__ int redi[kWavesPerBlock][kWave * 4];
  const int b = blockIdx.x / a.slices, s = blockIdx.x % a.slices;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int lpr = 1 << a.lpr_log2, groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2, cl = lane & (lpr - 1);
  const int beg = a.ptr[b], end = a.ptr[b + 1];
  const int len = (end - beg + a.slices - 1) / a.slices;
  const int r0 = beg + s * len, r1 = min(end, r0 + len);
  const bool is_max = a.kind == 2;
  for (int cbase = 0; cbase < a.d; cbase += lpr * 4) {
    const int c0 = cbase + cl * 4;
    const bool cact = c0 < a.d;
    float acc[4];
    int arg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = is_max ? kPoolNegBig : 0.f; arg[i] = -1; }
    for (int r = r0 + wave * groups + sub; r < r1; r += kWavesPerBlock * groups) {
      float v[4] = {0, 0, 0, 0};
      if (cact) load_vec<4>(v, a.x + (size_t)r * a.d + c0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (is_max) { if (v[i] > acc[i]) { acc[i] = v[i]; arg[i] = r; } }
        else acc[i] += v[i];
      }
    }
    // lane groups, then waves: larger value wins, on a tie the smaller row index
    for (int off = lpr; off < kWave; off <<= 1)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ov = __shfl_xor(acc[i], off);
        const int oa = __shfl_xor(arg[i], off);
        if (is_max) {
          if (oa >= 0 && (arg[i] < 0 || ov > acc[i] || (ov == acc[i] && oa < arg[i]))) { acc[i] = ov; arg[i] = oa; }
        } else acc[i] += ov;
      }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { redv[wave][cl * 4 + i] = acc[i]; redi[wave][cl * 4 + i] = arg[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < lpr * 4; c += kBlock) {
      if (cbase + c >= a.d) continue;
      float v = redv[0][c];
      int g = redi[0][c];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) {
        const float ov = redv[w][c];
        const int oa = redi[w][c];
        if (is_max) { if (oa >= 0 && (g < 0 || ov > v || (ov == v && oa < g))) { v = ov; g = oa; } }
        else v += ov;
      }
      const size_t o = ((size_t)b * a.slices + s) * a.d + cbase + c;
      a.part[o] = v;
      if (is_max) a.part_arg[o] = g;
    }
  }
}

__global__ __launch_bounds__(kBlock) void segment_pool_stage2_kernel(const PoolArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * a.d) return;
  const int b = idx / a.d, c = idx % a.d;
  const bool is_max = a.kind == 2;
  float v = is_max ? kPoolNegBig : 0.f;
  int g = -1;
  for (int s = 0; s < a.slices; ++s) {
    const size_t o = ((size_t)b * a.slices + s) * a.d + c;
    if (is_max) { const int oa = a.part_arg[o]; if (oa >= 0 && (g < 0 || a.part[o] > v)) { v = a.part[o]; g = oa; } }
    else v += a.part[o];
  }
  const int n = a.ptr[b + 1] - a.ptr[b];
  if (is_max) { a.out[idx] = g >= 0 ? v : 0.f; a.argmax[idx] = g; }
  else a.out[idx] = (a.kind == 1) ? v / (float)max(n, 1) : v;
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_segment_pool_workspace_bytes(int64_t B, int64_t d) {
  if (B < 0 || d <= 0) return MLGNN_E_SHAPE;
  return B * pool_slices(B) * d * 8;       // float partials + int32 argmax partials
}

extern "C" int mlgnn_segment_pool_fwd(const void* x, const int32_t* ptr, void* out, int32_t* argmax,
                                      void* workspace, int64_t workspace_bytes, int64_t B, int64_t d,
                                      int kind, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32) return MLGNN_E_DTYPE;
  if (B < 0 || d <= 0 || d % 4 != 0 || B * 8 > INT32_MAX) return MLGNN_E_SHAPE;
  const int slices = pool_slices(B);
  if (kind < 0 || kind > 2) return MLGNN_E_MODE;
  if (B == 0) return 0;
  if (!x || !ptr || !out || !workspace || (kind == 2 && !argmax)) return MLGNN_E_NULL;
  if (workspace_bytes < B * slices * d * 8) return MLGNN_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return MLGNN_E_ALIGN;
  PoolArgs a;
  a.x = (const float*)x; a.ptr = ptr; a.part = (float*)workspace;
  a.part_arg = (int*)((char*)workspace + (size_t)B * slices * d * 4);
  a.out = (float*)out; a.argmax = argmax; a.B = (int)B; a.d = (int)d; a.kind = kind; a.slices = slices;
  a.lpr_log2 = lanes_per_row_log2(d, 4);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(segment_pool_stage1_kernel, dim3((unsigned)(B * slices)), dim3(kBlock), 0, s, a);
  int err = (int)hipGetLastError();
  if (err) return err;
  hipLaunchKernelGGL(segment_pool_stage2_kernel, dim3((unsigned)((B * d + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
  return (int)hipGetLastError();
}
