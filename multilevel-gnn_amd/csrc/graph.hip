// COO -> CSR on the device: by-destination and by-source orderings of a batch's edge list.
//
// Reference: there is no counterpart -- PyG's MessagePassing.propagate re-gathers from the COO
// edge_index in every layer (models/gcn_lib/sparse/torch_vertex.py:82,277).  Here the topology is
// sorted once per batch with two stable LSD radix sorts over ceil(log2 N) key bits (rocPRIM
// onesweep through hipCUB): (dst, position) -> by-destination order, then (src, position) ->
// by-source order.  Stability keeps the COO order inside a row, which is what makes "first
// maximal edge wins" match torch_scatter's CPU loop.  Everything is enqueued on the caller's
// stream; no host synchronisation, workspace from the caller.
#include <hipcub/hipcub.hpp>
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

// node ids outside [0, N) are clamped (the aggregation kernels must never read out of bounds) and
// counted in *bad, which the host may inspect later (CSRGraph.validate) without a sync here
__device__ __forceinline__ int checked_id(int64_t v, int N, int* bad) {
  if (v < 0 || v >= N) {
    if (bad) atomicAdd(bad, 1);
    return v < 0 ? 0 : N - 1;
  }
  return (int)v;
}

__global__ void csr_prepare_kernel(const int64_t* __restrict__ key64, int* __restrict__ key32,
                                   int* __restrict__ iota, int64_t n, int N, int* bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { key32[i] = checked_id(key64[i], N, bad); iota[i] = (int)i; }
}

// rowptr[i] = first position whose sorted key is >= i  (keys sorted ascending, length n, rows N)
__global__ void csr_rowptr_kernel(const int* __restrict__ keys, int* __restrict__ rowptr, int64_t n, int N) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e > n) return;
  const int prev = (e == 0) ? -1 : keys[e - 1];
  const int cur = (e == n) ? N : keys[e];
  for (int i = prev + 1; i <= cur; ++i) rowptr[i] = (int)e;
}

__global__ void gather_i64_to_i32_kernel(const int64_t* __restrict__ src, const int* __restrict__ idx,
                                         int* __restrict__ out, int64_t n, int N, int* bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = checked_id(src[idx[i]], N, bad);
}

__global__ void iota_kernel(int* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int)i;
}

__global__ void gather2_kernel(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ idx,
                               int* __restrict__ out_a, int* __restrict__ out_b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int j = idx[i]; out_a[i] = a[j]; out_b[i] = b[j]; }
}

// attr [E0, r] (row stride `stride` floats, COO order) -> by_dst / by_src [E, width]: row e of the by-destination
// table is attr[eid[e]], of the by-source table attr[eid_t[e]], zero padded to `width` columns
__global__ void edge_table_kernel(const float* __restrict__ attr, int64_t stride, int r, int width,
                                  const int* __restrict__ eid, const int* __restrict__ eid_t,
                                  float* __restrict__ by_dst, float* __restrict__ by_src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* a = attr + (int64_t)eid[i] * stride;
  const float* b = attr + (int64_t)eid_t[i] * stride;
  for (int k = 0; k < width; ++k) {
    by_dst[i * width + k] = k < r ? a[k] : 0.f;
    by_src[i * width + k] = k < r ? b[k] : 0.f;
  }
}

static int key_bits(int64_t N) {
  int b = 1;
  while (((int64_t)1 << b) < N && b < 31) ++b;
  return b;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int64_t E, int bits) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr,
                                     (int*)nullptr, (int)E, 0, bits, (hipStream_t)0);
  return bytes;
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_coo_to_csr_workspace_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0 || N > INT32_MAX || E > INT32_MAX) return MLGNN_E_SHAPE;
  const size_t e4 = align256((size_t)E * 4);
  return (int64_t)(3 * e4 + align256(sort_temp_bytes(E, key_bits(N))) + 256);
}

extern "C" int mlgnn_coo_to_csr(const int64_t* edge_index, int64_t E, int64_t N,
                                int32_t* rowptr, int32_t* col, int32_t* eid,
                                int32_t* rowptr_t, int32_t* col_t, int32_t* pos_t, int32_t* eid_t,
                                int32_t* bad_ids, void* workspace, int64_t workspace_bytes, void* stream) {
  if (N < 0 || E < 0 || N > INT32_MAX || E > INT32_MAX) return MLGNN_E_SHAPE;
  if (!rowptr || !rowptr_t) return MLGNN_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int threads = 256;
  const unsigned gE1 = (unsigned)((E + 1 + threads - 1) / threads);
  if (bad_ids) (void)hipMemsetAsync(bad_ids, 0, 4, s);
  if (E == 0) {
    if (N >= 0) {
      (void)hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 4, s);
      (void)hipMemsetAsync(rowptr_t, 0, (size_t)(N + 1) * 4, s);
    }
    return (int)hipGetLastError();
  }
  if (!edge_index || !col || !eid || !col_t || !pos_t || !eid_t || !workspace) return MLGNN_E_NULL;
  const int bits = key_bits(N);
  const size_t e4 = align256((size_t)E * 4);
  size_t temp_bytes = sort_temp_bytes(E, bits);
  if (workspace_bytes < (int64_t)(3 * e4 + align256(temp_bytes) + 256)) return MLGNN_E_WORKSPACE;
  char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  int* key_in = (int*)w;
  int* val_in = (int*)(w + e4);
  int* key_out = (int*)(w + 2 * e4);          // sorted keys: dst (pass 1), src (pass 2)
  void* temp = w + 3 * e4;
  const unsigned gE = (unsigned)((E + threads - 1) / threads);
  const int64_t* src64 = edge_index;
  const int64_t* dst64 = edge_index + E;

  // ---- by destination: stable sort of (dst, position) -------------------------------------------
  hipLaunchKernelGGL(csr_prepare_kernel, dim3(gE), dim3(threads), 0, s, dst64, key_in, val_in, E, (int)N, bad_ids);
  hipError_t err = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key_in, key_out, val_in, eid, (int)E, 0, bits, s);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3(gE1), dim3(threads), 0, s, key_out, rowptr, E, (int)N);
  hipLaunchKernelGGL(gather_i64_to_i32_kernel, dim3(gE), dim3(threads), 0, s, src64, eid, col, E, (int)N, bad_ids);
  // dst in by-destination order stays in key_out until the second sort has consumed `col`;
  // copy it aside into key_in (free now) because the second sort overwrites key_out
  (void)hipMemcpyAsync(key_in, key_out, (size_t)E * 4, hipMemcpyDeviceToDevice, s);

  // ---- by source: stable sort of (src of the by-destination order, by-destination position) -----
  hipLaunchKernelGGL(iota_kernel, dim3(gE), dim3(threads), 0, s, val_in, E);
  err = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, col, key_out, val_in, pos_t, (int)E, 0, bits, s);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3(gE1), dim3(threads), 0, s, key_out, rowptr_t, E, (int)N);
  hipLaunchKernelGGL(gather2_kernel, dim3(gE), dim3(threads), 0, s, key_in, eid, pos_t, col_t, eid_t, E);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_edge_table_to_csr(const float* attr, int64_t row_stride, int64_t r, int64_t width,
                                       const int32_t* eid, const int32_t* eid_t, float* by_dst, float* by_src,
                                       int64_t E, void* stream) {
  if (E < 0 || r < 1 || width < r || width > 8 || row_stride < r) return MLGNN_E_SHAPE;
  if (E == 0) return 0;
  if (!attr || !eid || !eid_t || !by_dst || !by_src) return MLGNN_E_NULL;
  const int threads = 256;
  hipLaunchKernelGGL(edge_table_kernel, dim3((unsigned)((E + threads - 1) / threads)), dim3(threads), 0,
                     (hipStream_t)stream, attr, row_stride, (int)r, (int)width, eid, eid_t, by_dst, by_src, E);
  return (int)hipGetLastError();
}
