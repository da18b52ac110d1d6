// COO -> CSR on the device: by-destination and by-source orderings of a batch's edge list.
//
// Reference: there is no counterpart -- PyG's MessagePassing.propagate re-gathers from the COO
// edge_index in every layer (models/gcn_lib/sparse/torch_vertex.py:82,277).  Here the topology is
// ordered once per batch.  Rows are short (mean degree 16), so instead of two full-width stable radix
// sorts (six passes over all edges) the build is a counting sort with a per-row fix-up:
//   1. degree histograms of dst and src (atomic adds), exclusive scans -> rowptr, rowptr_t;
//   2. every edge takes the next free slot of its destination row (atomic cursor: arrival order);
//   3. each row is sorted by COO position -- two rows per wavefront with a 32-lane bitonic network on
//      shuffles, one row per wavefront up to 64 edges, longer rows in a second launch (LDS / global bitonic)
//      -- which restores the stable order: "first maximal edge wins" then matches torch_scatter's CPU loop,
//      and the result is the same on every run whatever the arrival order was;
//   4. the same for the source rows, keyed by the by-destination position.
// Everything is enqueued on the caller's stream; no host synchronisation, workspace from the caller.
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

// The histogram and placement passes keep their atomics in LDS.  A workgroup takes a chunk of kCsrChunk
// consecutive edges; in a block-diagonal batch their endpoints fall into one graph, i.e. into a window of a few
// thousand consecutive rows, so the workgroup counts into an LDS window of kCsrWindow rows starting at the
// chunk's smallest id and touches global memory with one COALESCED atomic per window row (a scattered global
// atomic per edge runs at ~2.5e10/s on this chip: 0.75 ms for the two histograms of 10 M edges).  Ids outside
// the window (unsorted edge lists) fall back to per-edge global atomics: slower, same result.
constexpr int kCsrThreads = 1024, kCsrPerThread = 32, kCsrChunk = kCsrThreads * kCsrPerThread;
constexpr int kCsrWindow = 16384;

__device__ __forceinline__ int block_min(int v, int* slot) {
  if (threadIdx.x == 0) *slot = INT32_MAX;
  __syncthreads();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
  if ((threadIdx.x & (kWave - 1)) == 0) atomicMin(slot, v);
  __syncthreads();
  return *slot;
}

// pass 1: clamp + narrow the endpoints, degree histograms of both
__global__ __launch_bounds__(kCsrThreads) void csr_count_kernel(const int64_t* __restrict__ src64,
                                                                const int64_t* __restrict__ dst64,
                                                                int* __restrict__ src32, int* __restrict__ dst32,
                                                                int* __restrict__ cnt_dst, int* __restrict__ cnt_src,
                                                                int64_t n, int N, int* bad) {
  __shared__ int hd[kCsrWindow], hs[kCsrWindow];
  __shared__ int slot_d, slot_s;
  const int64_t e0 = (int64_t)blockIdx.x * kCsrChunk;
  int kd[kCsrPerThread], ks[kCsrPerThread];
  int md = INT32_MAX, ms = INT32_MAX;
#pragma unroll
  for (int q = 0; q < kCsrPerThread; ++q) {
    const int64_t i = e0 + threadIdx.x + (int64_t)q * kCsrThreads;
    kd[q] = -1; ks[q] = -1;
    if (i < n) {
      ks[q] = checked_id(src64[i], N, bad); kd[q] = checked_id(dst64[i], N, bad);
      src32[i] = ks[q]; dst32[i] = kd[q];
      md = min(md, kd[q]); ms = min(ms, ks[q]);
    }
  }
  for (int i = threadIdx.x; i < kCsrWindow; i += kCsrThreads) { hd[i] = 0; hs[i] = 0; }
  const int base_d = block_min(md, &slot_d), base_s = block_min(ms, &slot_s);     // (barriers inside)
#pragma unroll
  for (int q = 0; q < kCsrPerThread; ++q) {
    if (kd[q] < 0) continue;
    const int od = kd[q] - base_d, os = ks[q] - base_s;
    if (od < kCsrWindow) atomicAdd(hd + od, 1); else atomicAdd(cnt_dst + kd[q], 1);
    if (os < kCsrWindow) atomicAdd(hs + os, 1); else atomicAdd(cnt_src + ks[q], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCsrWindow; i += kCsrThreads) {
    if (hd[i]) atomicAdd(cnt_dst + base_d + i, hd[i]);
    if (hs[i]) atomicAdd(cnt_src + base_s + i, hs[i]);
  }
}

// pass 2: element i (value = i) goes to the next free slot of row key[i].  The workgroup reserves, per window
// row, a contiguous range of the row with ONE global atomic and hands its slots out with LDS atomics.
__global__ __launch_bounds__(kCsrThreads) void csr_place_kernel(const int* __restrict__ key,
                                                                const int* __restrict__ rowptr,
                                                                int* __restrict__ cursor, int* __restrict__ out,
                                                                int64_t n) {
  __shared__ int h[kCsrWindow], hb[kCsrWindow];
  __shared__ int slot;
  const int64_t e0 = (int64_t)blockIdx.x * kCsrChunk;
  int k[kCsrPerThread];
  int mk = INT32_MAX;
#pragma unroll
  for (int q = 0; q < kCsrPerThread; ++q) {
    const int64_t i = e0 + threadIdx.x + (int64_t)q * kCsrThreads;
    k[q] = i < n ? key[i] : -1;
    if (k[q] >= 0) mk = min(mk, k[q]);
  }
  for (int i = threadIdx.x; i < kCsrWindow; i += kCsrThreads) h[i] = 0;
  const int base = block_min(mk, &slot);
#pragma unroll
  for (int q = 0; q < kCsrPerThread; ++q)
    if (k[q] >= 0 && k[q] - base < kCsrWindow) atomicAdd(h + (k[q] - base), 1);
  __syncthreads();
  for (int i = threadIdx.x; i < kCsrWindow; i += kCsrThreads) {
    const int c = h[i];
    if (c) hb[i] = rowptr[base + i] + atomicAdd(cursor + base + i, c);
    h[i] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kCsrPerThread; ++q) {
    if (k[q] < 0) continue;
    const int64_t i = e0 + threadIdx.x + (int64_t)q * kCsrThreads;
    const int o = k[q] - base;
    const int where = o < kCsrWindow ? hb[o] + atomicAdd(h + o, 1) : rowptr[k[q]] + atomicAdd(cursor + k[q], 1);
    out[where] = (int)i;
  }
}

// ascending bitonic network on the lanes of a wavefront, all comparators pointing the same way (first step of
// every merge mirrors inside its block), so INT_MAX padding at the top never moves.  WIDTH = 32 or 64 lanes.
template <int WIDTH>
__device__ __forceinline__ int wave_sort(int v, int lane) {
#pragma unroll
  for (int k = 2; k <= WIDTH; k <<= 1) {
    int p = lane ^ (k - 1);
    int o = __shfl(v, p);
    v = (lane < p) ? min(v, o) : max(v, o);
#pragma unroll
    for (int j = k >> 2; j >= 1; j >>= 1) {
      p = lane ^ j;
      o = __shfl(v, p);
      v = (lane < p) ? min(v, o) : max(v, o);
    }
  }
  return v;
}

// What a sorted row slot also writes.  by destination (BY_SRC = false): data[p] = COO position e of the edge;
// col[p] = src[e], rowkey[p] = its destination row.  by source (BY_SRC = true): data[q] = by-destination
// position p of the edge; col_t[q] = rowkey[p] (its destination), eid_t[q] = eid[p].
struct RowOut { const int* src32; int* col; int* rowkey; const int* eid; int* col_t; int* eid_t; };

template <bool BY_SRC>
__device__ __forceinline__ void emit(const RowOut& o, int slot, int v, int row) {
  if constexpr (!BY_SRC) { o.col[slot] = o.src32[v]; o.rowkey[slot] = row; }
  else { o.col_t[slot] = o.rowkey[v]; o.eid_t[slot] = o.eid[v]; }
}

// pass 3: sort every row of `data` (unique keys) in place and emit the dependent arrays; rows longer than 64
// are appended to long_rows
template <bool BY_SRC>
__global__ __launch_bounds__(kBlock) void csr_sort_rows_kernel(const int* __restrict__ rowptr, int* __restrict__ data,
                                                               int N, int* __restrict__ long_rows,
                                                               int* __restrict__ long_count, const RowOut o) {
  const int lane = threadIdx.x & (kWave - 1);
  const int pair = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int r0 = 2 * pair;
  if (r0 >= N) return;
  const int b0 = rowptr[r0], e0 = rowptr[r0 + 1];
  const int e1 = (r0 + 1 < N) ? rowptr[r0 + 2] : e0;
  const int d0 = e0 - b0, d1 = e1 - e0;
  if (max(d0, d1) <= 32) {                       // two rows per wavefront, one per 32-lane half
    const int half = lane >> 5, l = lane & 31;
    const int beg = half ? e0 : b0, deg = half ? d1 : d0;
    int v = l < deg ? data[beg + l] : INT32_MAX;
    v = wave_sort<32>(v, lane);                  // partners stay inside the half (xor with < 32)
    if (l < deg) { data[beg + l] = v; emit<BY_SRC>(o, beg + l, v, r0 + half); }
    return;
  }
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const int beg = which ? e0 : b0, deg = which ? d1 : d0;
    if (deg <= kWave) {
      int v = lane < deg ? data[beg + lane] : INT32_MAX;
      v = wave_sort<kWave>(v, lane);
      if (lane < deg) { data[beg + lane] = v; emit<BY_SRC>(o, beg + lane, v, r0 + which); }
    } else if (lane == 0) {
      long_rows[atomicAdd(long_count, 1)] = r0 + which;
    }
  }
}

// pass 3b: rows of more than 64 edges, one workgroup each: bitonic network in LDS (<= 4096) or in global memory
constexpr int kLongLds = 4096;
template <bool BY_SRC>
__global__ __launch_bounds__(kBlock) void csr_sort_long_rows_kernel(const int* __restrict__ rowptr,
                                                                    int* __restrict__ data,
                                                                    const int* __restrict__ long_rows,
                                                                    const int* __restrict__ long_count,
                                                                    const RowOut o) {
  __shared__ int buf[kLongLds];
  const int count = *long_count;
  for (int it = blockIdx.x; it < count; it += gridDim.x) {
    const int r = long_rows[it];
    const int beg = rowptr[r], n = rowptr[r + 1] - beg;
    int P = 1;
    while (P < n) P <<= 1;
    const bool in_lds = n <= kLongLds;
    int* a = in_lds ? buf : data + beg;
    if (in_lds) {
      for (int i = threadIdx.x; i < n; i += kBlock) buf[i] = data[beg + i];
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
      for (int i = threadIdx.x; i < P / 2; i += kBlock) {          // mirror step
        const int blk = i / (k / 2), off = i % (k / 2);
        const int x = blk * k + off, y = blk * k + k - 1 - off;
        if (y < n) { const int u = a[x], w = a[y]; if (u > w) { a[x] = w; a[y] = u; } }
      }
      __syncthreads();
      for (int j = k >> 2; j >= 1; j >>= 1) {
        for (int i = threadIdx.x; i < P / 2; i += kBlock) {
          const int x = (i / j) * 2 * j + i % j, y = x + j;
          if (y < n) { const int u = a[x], w = a[y]; if (u > w) { a[x] = w; a[y] = u; } }
        }
        __syncthreads();
      }
    }
    for (int i = threadIdx.x; i < n; i += kBlock) {
      const int v = a[i];
      if (in_lds) data[beg + i] = v;
      emit<BY_SRC>(o, beg + i, v, r);
    }
    __syncthreads();
  }
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

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t scan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (int)n, (hipStream_t)0);
  return bytes;
}

// workspace layout (256-byte aligned pieces)
struct CsrWs {
  int *src32, *dst32, *rowkey, *cnt_dst, *cnt_src, *long_rows, *long_count;
  void* scan_temp; size_t scan_bytes; size_t total;
};
static CsrWs csr_ws(void* workspace, int64_t N, int64_t E) {
  CsrWs w;
  const size_t e4 = align256((size_t)E * 4), n4 = align256((size_t)(N + 1) * 4);
  char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  w.src32 = (int*)p; p += e4;
  w.dst32 = (int*)p; p += e4;
  w.rowkey = (int*)p; p += e4;                  // destination row of every by-destination slot
  w.cnt_dst = (int*)p; p += n4;                 // histogram, then reused as the placement cursor
  w.cnt_src = (int*)p; p += n4;
  w.long_rows = (int*)p; p += n4;
  w.long_count = (int*)p; p += 256;
  w.scan_bytes = scan_temp_bytes(N + 1);
  w.scan_temp = p; p += align256(w.scan_bytes);
  w.total = (size_t)(p - (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255)) + 256;
  return w;
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_coo_to_csr_workspace_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0 || N > INT32_MAX - 1 || E > INT32_MAX) return MLGNN_E_SHAPE;
  return (int64_t)csr_ws(nullptr, N, E).total;
}

extern "C" int mlgnn_coo_to_csr(const int64_t* edge_index, int64_t E, int64_t N,
                                int32_t* rowptr, int32_t* col, int32_t* eid,
                                int32_t* rowptr_t, int32_t* col_t, int32_t* pos_t, int32_t* eid_t,
                                int32_t* bad_ids, void* workspace, int64_t workspace_bytes, void* stream) {
  if (N < 0 || E < 0 || N > INT32_MAX - 1 || E > INT32_MAX) return MLGNN_E_SHAPE;
  if (!rowptr || !rowptr_t) return MLGNN_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int threads = 256;
  if (bad_ids) (void)hipMemsetAsync(bad_ids, 0, 4, s);
  if (E == 0) {
    (void)hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 4, s);
    (void)hipMemsetAsync(rowptr_t, 0, (size_t)(N + 1) * 4, s);
    return (int)hipGetLastError();
  }
  if (!edge_index || !col || !eid || !col_t || !pos_t || !eid_t || !workspace) return MLGNN_E_NULL;
  if (N == 0) return MLGNN_E_SHAPE;                               // edges without nodes
  const CsrWs w = csr_ws(workspace, N, E);
  if (workspace_bytes < (int64_t)w.total) return MLGNN_E_WORKSPACE;
  const unsigned gC = (unsigned)((E + kCsrChunk - 1) / kCsrChunk);
  const unsigned gRows = (unsigned)(((N + 1) / 2 + kWavesPerBlock - 1) / kWavesPerBlock);
  const size_t n4 = (size_t)(N + 1) * 4;
  hipError_t err;
  (void)threads;

  // ---- degrees and row pointers ---------------------------------------------------------------------
  (void)hipMemsetAsync(w.cnt_dst, 0, (size_t)((char*)w.long_count - (char*)w.cnt_dst) + 4, s);   // cnt_dst .. long_count
  hipLaunchKernelGGL(csr_count_kernel, dim3(gC), dim3(kCsrThreads), 0, s, edge_index, edge_index + E, w.src32,
                     w.dst32, w.cnt_dst, w.cnt_src, E, (int)N, bad_ids);
  size_t sb = w.scan_bytes;
  err = hipcub::DeviceScan::ExclusiveSum(w.scan_temp, sb, w.cnt_dst, rowptr, (int)(N + 1), s);
  if (err != hipSuccess) return (int)err;
  err = hipcub::DeviceScan::ExclusiveSum(w.scan_temp, sb, w.cnt_src, rowptr_t, (int)(N + 1), s);
  if (err != hipSuccess) return (int)err;

  RowOut o;
  o.src32 = w.src32; o.col = col; o.rowkey = w.rowkey; o.eid = eid; o.col_t = col_t; o.eid_t = eid_t;
  // ---- by destination: slot by arrival, rows sorted by COO position ---------------------------------------
  (void)hipMemsetAsync(w.cnt_dst, 0, n4, s);
  hipLaunchKernelGGL(csr_place_kernel, dim3(gC), dim3(kCsrThreads), 0, s, w.dst32, rowptr, w.cnt_dst, eid, E);
  hipLaunchKernelGGL(csr_sort_rows_kernel<false>, dim3(gRows), dim3(kBlock), 0, s, rowptr, eid, (int)N, w.long_rows,
                     w.long_count, o);
  hipLaunchKernelGGL(csr_sort_long_rows_kernel<false>, dim3(256), dim3(kBlock), 0, s, rowptr, eid, w.long_rows,
                     w.long_count, o);

  // ---- by source: the by-destination slots p = 0..E-1 keyed by their source col[p], rows sorted by p ---------
  (void)hipMemsetAsync(w.cnt_src, 0, n4, s);
  (void)hipMemsetAsync(w.long_count, 0, 4, s);
  hipLaunchKernelGGL(csr_place_kernel, dim3(gC), dim3(kCsrThreads), 0, s, col, rowptr_t, w.cnt_src, pos_t, E);
  hipLaunchKernelGGL(csr_sort_rows_kernel<true>, dim3(gRows), dim3(kBlock), 0, s, rowptr_t, pos_t, (int)N, w.long_rows,
                     w.long_count, o);
  hipLaunchKernelGGL(csr_sort_long_rows_kernel<true>, dim3(256), dim3(kBlock), 0, s, rowptr_t, pos_t, w.long_rows,
                     w.long_count, o);
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
