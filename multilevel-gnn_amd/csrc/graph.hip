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

// The histogram and placement passes keep their atomics in LDS.  A workgroup takes a chunk of `per` * 1024
// consecutive edges; in a block-diagonal batch their endpoints fall into one graph, i.e. into a window of a few
// thousand consecutive rows, so the workgroup counts into an LDS window of kCsrWindow rows starting at the
// chunk's smallest id and touches global memory with one COALESCED atomic per window row (a scattered global
// atomic per edge runs at ~2.5e10/s on this chip: 0.75 ms for the two histograms of 10 M edges).  Ids outside
// the window (unsorted edge lists) fall back to per-edge global atomics: slower, same result.
//
// Launch shape (round 3): ONE 64 KB window per workgroup and <= 64 VGPRs, so two 1024-thread workgroups share a CU;
// the chunk length is chosen on the host so that the whole edge list is one resident round of about 2 x 256
// workgroups (the earlier fixed 32768-edge chunks gave 312 workgroups at one per CU: a full round and a 22 % one).
// Chunks are dealt to the XCDs in contiguous runs (workgroups b, b + 8, ... share an XCD and its L2): the scattered
// 4-byte writes of the placement then fall, per XCD, into the 640 KB window of the graph that XCD is working on, and
// a cache line is completed by ONE L2 before it is written back (with interleaved chunks every line of `out` was
// assembled from partial writes of up to eight L2s: 5 x the bytes of the array reached HBM).
constexpr int kCsrThreads = 1024, kCsrMaxPer = 20;
constexpr int kCsrWindow = 16384;
constexpr int kCsrTargetBlocks = 512;                  // two per CU

// chunk a workgroup works on: XCD x owns chunks [x * nb / 8, (x + 1) * nb / 8) in order
__device__ __forceinline__ int csr_chunk_of_block() {
  const int nb = gridDim.x, q = nb / kXcds, r = nb % kXcds, xcd = blockIdx.x % kXcds;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + blockIdx.x / kXcds;
}

__device__ __forceinline__ int block_min(int v, int* slot) {
  if (threadIdx.x == 0) *slot = INT32_MAX;
  __syncthreads();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
  if ((threadIdx.x & (kWave - 1)) == 0) atomicMin(slot, v);
  __syncthreads();
  return *slot;
}

// degree histogram of one endpoint list of the chunk: clamp + narrow to int32, count in the LDS window, flush
__device__ __forceinline__ void csr_count_side(const int64_t* __restrict__ in64, int* __restrict__ out32,
                                               int* __restrict__ cnt, int64_t e0, int64_t n, int per, int N, int* bad,
                                               int* h, int* slot) {
  int k[kCsrMaxPer];
  int mk = INT32_MAX;
  // (64-bit loads in groups of 5: twenty of them in flight would take 40 of the 64 registers)
  const int64_t first = e0 + threadIdx.x;
  const int64_t left = n - first;
  const int lim = (int)(left < (int64_t)per * kCsrThreads ? (left < 0 ? 0 : left) : (int64_t)per * kCsrThreads);
  const int64_t* __restrict__ in = in64 + first;
  int* __restrict__ out = out32 + first;
#pragma unroll
  for (int g = 0; g < kCsrMaxPer; g += 5) {
    int64_t raw[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) raw[q] = ((g + q) * kCsrThreads < lim) ? in[(g + q) * kCsrThreads] : -1;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      k[g + q] = -1;
      if ((g + q) * kCsrThreads < lim) {
        k[g + q] = checked_id(raw[q], N, bad);
        out[(g + q) * kCsrThreads] = k[g + q];
        mk = min(mk, k[g + q]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int i = threadIdx.x; i < kCsrWindow / 4; i += kCsrThreads) reinterpret_cast<int4*>(h)[i] = make_int4(0, 0, 0, 0);
  const int base = block_min(mk, slot);                                           // (barriers inside)
#pragma unroll
  for (int q = 0; q < kCsrMaxPer; ++q) {
    if (q >= per || k[q] < 0) continue;
    const int o = k[q] - base;
    if (o < kCsrWindow) atomicAdd(h + o, 1); else atomicAdd(cnt + k[q], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCsrWindow; i += kCsrThreads)
    if (h[i]) atomicAdd(cnt + base + i, h[i]);
  __syncthreads();
}

// pass 1: clamp + narrow the endpoints, degree histograms of both (one after the other through the same window)
__global__ __launch_bounds__(kCsrThreads, 8) void csr_count_kernel(const int64_t* __restrict__ src64,
                                                                   const int64_t* __restrict__ dst64,
                                                                   int* __restrict__ src32, int* __restrict__ dst32,
                                                                   int* __restrict__ cnt_dst, int* __restrict__ cnt_src,
                                                                   int64_t n, int per, int N, int* bad) {
  __shared__ __attribute__((aligned(16))) int h[kCsrWindow];
  __shared__ int slot;
  const int64_t e0 = (int64_t)csr_chunk_of_block() * per * kCsrThreads;
  csr_count_side(dst64, dst32, cnt_dst, e0, n, per, N, bad, h, &slot);
  csr_count_side(src64, src32, cnt_src, e0, n, per, N, bad, h, &slot);
}

// pass 2: element i (value = i) goes to the next free slot of row key[i].  The workgroup reserves, per window
// row, a contiguous range of the row with ONE global atomic and hands its slots out with LDS atomics: h[] holds the
// chunk's count of the row, then the next free slot of the reserved range.
__global__ __launch_bounds__(kCsrThreads, 8) void csr_place_kernel(const int* __restrict__ key,
                                                                   const int* __restrict__ rowptr,
                                                                   int* __restrict__ cursor, int* __restrict__ out,
                                                                   int64_t n, int per) {
  __shared__ __attribute__((aligned(16))) int h[kCsrWindow];
  __shared__ int slot;
  const int64_t e0 = (int64_t)csr_chunk_of_block() * per * kCsrThreads;
  int k[kCsrMaxPer];
  int mk = INT32_MAX;
#pragma unroll
  for (int q = 0; q < kCsrMaxPer; ++q) {
    const int64_t i = e0 + threadIdx.x + (int64_t)q * kCsrThreads;
    k[q] = (q < per && i < n) ? key[i] : -1;
    if (k[q] >= 0) mk = min(mk, k[q]);
  }
  for (int i = threadIdx.x; i < kCsrWindow / 4; i += kCsrThreads) reinterpret_cast<int4*>(h)[i] = make_int4(0, 0, 0, 0);
  const int base = block_min(mk, &slot);
#pragma unroll
  for (int q = 0; q < kCsrMaxPer; ++q)
    if (q < per && k[q] >= 0 && k[q] - base < kCsrWindow) atomicAdd(h + (k[q] - base), 1);
  __syncthreads();
  for (int i = threadIdx.x; i < kCsrWindow; i += kCsrThreads) {
    const int c = h[i];
    if (c) h[i] = rowptr[base + i] + atomicAdd(cursor + base + i, c);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kCsrMaxPer; ++q) {
    if (q >= per || k[q] < 0) continue;
    const int64_t i = e0 + threadIdx.x + (int64_t)q * kCsrThreads;
    const int o = k[q] - base;
    const int where = o < kCsrWindow ? atomicAdd(h + o, 1) : rowptr[k[q]] + atomicAdd(cursor + k[q], 1);
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
// are appended to long_rows.  A wavefront takes 2 * kSortPairs consecutive rows: their row pointers are one load, and
// when all of them hold at most 32 edges (two rows per wavefront, one per 32-lane half) the loads of the pairs, their
// sorting networks and the gathers of the dependent arrays are issued pair after pair before the first result is
// needed -- the kernel is a chain of three dependent memory round trips per row (row pointer -> keys -> gathered
// values) and ran 2.5 x over its traffic bound with one pair per wavefront.
constexpr int kSortPairs = 4;
constexpr int kSortRows = 2 * kSortPairs;

template <bool BY_SRC>
__global__ __launch_bounds__(kBlock) void csr_sort_rows_kernel(const int* __restrict__ rowptr, int* __restrict__ data,
                                                               int N, int* __restrict__ long_rows,
                                                               int* __restrict__ long_count, const RowOut o) {
  const int lane = threadIdx.x & (kWave - 1);
  const int r0 = (blockIdx.x * kWavesPerBlock + threadIdx.x / kWave) * kSortRows;
  if (r0 >= N) return;
  const int rp = rowptr[min(r0 + min(lane, kSortRows), N)];          // rows past N: empty
  const int deg_l = __shfl_down(rp, 1) - rp;
  const bool small = __builtin_amdgcn_ballot_w64(lane < kSortRows && deg_l > 32) == 0;
  if (small) {
    const int half = lane >> 5, l = lane & 31;
    int v[kSortPairs], beg[kSortPairs], dg[kSortPairs];
#pragma unroll
    for (int p = 0; p < kSortPairs; ++p) {
      beg[p] = __shfl(rp, 2 * p + half);
      dg[p] = __shfl(rp, 2 * p + half + 1) - beg[p];
      v[p] = l < dg[p] ? data[beg[p] + l] : INT32_MAX;
    }
#pragma unroll
    for (int p = 0; p < kSortPairs; ++p) v[p] = wave_sort<32>(v[p], lane);     // partners stay inside the half
    int g0[kSortPairs], g1[kSortPairs];
#pragma unroll
    for (int p = 0; p < kSortPairs; ++p) {
      g0[p] = 0; g1[p] = 0;
      if (l < dg[p]) {
        if constexpr (!BY_SRC) g0[p] = o.src32[v[p]];
        else { g0[p] = o.rowkey[v[p]]; g1[p] = o.eid[v[p]]; }
      }
    }
#pragma unroll
    for (int p = 0; p < kSortPairs; ++p) {
      if (l >= dg[p]) continue;
      const int slot = beg[p] + l;
      data[slot] = v[p];
      if constexpr (!BY_SRC) { o.col[slot] = g0[p]; o.rowkey[slot] = r0 + 2 * p + half; }
      else { o.col_t[slot] = g0[p]; o.eid_t[slot] = g1[p]; }
    }
    return;
  }
  for (int w = 0; w < kSortRows && r0 + w < N; ++w) {
    const int beg = __shfl(rp, w), deg = __shfl(rp, w + 1) - beg;
    if (deg <= kWave) {
      int v = lane < deg ? data[beg + lane] : INT32_MAX;
      v = wave_sort<kWave>(v, lane);
      if (lane < deg) { data[beg + lane] = v; emit<BY_SRC>(o, beg + lane, v, r0 + w); }
    } else if (lane == 0) {
      long_rows[atomicAdd(long_count, 1)] = r0 + w;
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

// one scalar per edge (the rank-1 edge term of BASELINE configs[1]): four edges per thread, 16-byte index loads and
// table stores, the eight gathers of a thread in flight together.  n4 = n / 4 groups; the tail runs on the kernel above.
__global__ __launch_bounds__(256) void edge_table_w1_kernel(const float* __restrict__ attr, int64_t stride,
                                                            const int4* __restrict__ eid, const int4* __restrict__ eid_t,
                                                            float4* __restrict__ by_dst, float4* __restrict__ by_src,
                                                            int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int4 e = eid[i], t = eid_t[i];
  float4 a, b;
  a.x = attr[(int64_t)e.x * stride]; a.y = attr[(int64_t)e.y * stride];
  a.z = attr[(int64_t)e.z * stride]; a.w = attr[(int64_t)e.w * stride];
  b.x = attr[(int64_t)t.x * stride]; b.y = attr[(int64_t)t.y * stride];
  b.z = attr[(int64_t)t.z * stride]; b.w = attr[(int64_t)t.w * stride];
  by_dst[i] = a;
  by_src[i] = b;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t scan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (int)n, (hipStream_t)0);
  return bytes;
}

// workspace layout (256-byte aligned pieces)
struct CsrWs {
  int *src32, *dst32, *rowkey, *cnt_dst, *cnt_src, *cur_dst, *cur_src, *long_rows, *long_rows_t, *long_count;
  size_t zero_bytes;                       // cnt_dst .. long_count: cleared by ONE fill at the start of a build
  void* scan_temp; size_t scan_bytes; size_t total;
};
static CsrWs csr_ws(void* workspace, int64_t N, int64_t E) {
  CsrWs w;
  const size_t e4 = align256((size_t)E * 4), n4 = align256((size_t)(N + 1) * 4);
  char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  w.src32 = (int*)p; p += e4;
  w.dst32 = (int*)p; p += e4;
  w.rowkey = (int*)p; p += e4;                  // destination row of every by-destination slot
  w.long_rows = (int*)p; p += n4;               // (written before they are read: not cleared)
  w.long_rows_t = (int*)p; p += n4;
  w.cnt_dst = (int*)p; p += n4;                 // degree histograms
  w.cnt_src = (int*)p; p += n4;
  w.cur_dst = (int*)p; p += n4;                 // placement cursors
  w.cur_src = (int*)p; p += n4;
  w.long_count = (int*)p; p += 256;             // [0]: by destination, [1]: by source
  w.zero_bytes = (size_t)(p - (char*)w.cnt_dst);
  w.scan_bytes = scan_temp_bytes(N + 1);
  w.scan_temp = p; p += align256(w.scan_bytes);
  w.total = (size_t)(p - (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255)) + 256;
  return w;
}

// edges per thread of the count / placement workgroups: the whole list in about kCsrTargetBlocks chunks
static int csr_per_thread(int64_t E) {
  const int64_t target = (E + kCsrTargetBlocks - 1) / kCsrTargetBlocks;
  const int64_t per = (target + kCsrThreads - 1) / kCsrThreads;
  return (int)(per < 1 ? 1 : per > kCsrMaxPer ? kCsrMaxPer : per);
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
  const int per = csr_per_thread(E);
  const int64_t chunk = (int64_t)per * kCsrThreads;
  const unsigned gC = (unsigned)((E + chunk - 1) / chunk);
  const unsigned gRows = (unsigned)(((N + kSortRows - 1) / kSortRows + kWavesPerBlock - 1) / kWavesPerBlock);
  hipError_t err;

  // ---- degrees and row pointers ---------------------------------------------------------------------
  (void)hipMemsetAsync(w.cnt_dst, 0, w.zero_bytes, s);            // histograms, cursors, long-row counters
  hipLaunchKernelGGL(csr_count_kernel, dim3(gC), dim3(kCsrThreads), 0, s, edge_index, edge_index + E, w.src32,
                     w.dst32, w.cnt_dst, w.cnt_src, E, per, (int)N, bad_ids);
  size_t sb = w.scan_bytes;
  err = hipcub::DeviceScan::ExclusiveSum(w.scan_temp, sb, w.cnt_dst, rowptr, (int)(N + 1), s);
  if (err != hipSuccess) return (int)err;
  err = hipcub::DeviceScan::ExclusiveSum(w.scan_temp, sb, w.cnt_src, rowptr_t, (int)(N + 1), s);
  if (err != hipSuccess) return (int)err;

  RowOut o;
  o.src32 = w.src32; o.col = col; o.rowkey = w.rowkey; o.eid = eid; o.col_t = col_t; o.eid_t = eid_t;
  // ---- by destination: slot by arrival, rows sorted by COO position ---------------------------------------
  hipLaunchKernelGGL(csr_place_kernel, dim3(gC), dim3(kCsrThreads), 0, s, w.dst32, rowptr, w.cur_dst, eid, E, per);
  hipLaunchKernelGGL(csr_sort_rows_kernel<false>, dim3(gRows), dim3(kBlock), 0, s, rowptr, eid, (int)N, w.long_rows,
                     w.long_count, o);
  hipLaunchKernelGGL(csr_sort_long_rows_kernel<false>, dim3(256), dim3(kBlock), 0, s, rowptr, eid, w.long_rows,
                     w.long_count, o);

  // ---- by source: the by-destination slots p = 0..E-1 keyed by their source col[p], rows sorted by p ---------
  hipLaunchKernelGGL(csr_place_kernel, dim3(gC), dim3(kCsrThreads), 0, s, col, rowptr_t, w.cur_src, pos_t, E, per);
  hipLaunchKernelGGL(csr_sort_rows_kernel<true>, dim3(gRows), dim3(kBlock), 0, s, rowptr_t, pos_t, (int)N, w.long_rows_t,
                     w.long_count + 1, o);
  hipLaunchKernelGGL(csr_sort_long_rows_kernel<true>, dim3(256), dim3(kBlock), 0, s, rowptr_t, pos_t, w.long_rows_t,
                     w.long_count + 1, o);
  return (int)hipGetLastError();
}

namespace mlgnn {
// SAGEConv's edge list (reference: models/gcn_lib/sparse/torch_vertex.py:272-273 `remove_self_loops` then
// `add_self_loops(..., fill_value=1.0)`) in ONE pass and without a compaction: an existing self loop (i, i) is parked on
// the spare node N -- a row nobody aggregates and nobody gathers from -- instead of being squeezed out of the list, and
// the N loops (i, i) with weight 1 follow the E original edges.  The CSR is then built over N + 1 nodes and used with N.
__global__ __launch_bounds__(256) void sage_rewrite_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                          const float* __restrict__ attr, int64_t attr_stride,
                                                          int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst,
                                                          float* __restrict__ out_w, int64_t E, int64_t N) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E + N; i += stride) {
    if (i < E) {
      const int64_t a = src[i], b = dst[i];
      const bool loop = a == b;
      out_src[i] = loop ? N : a;
      out_dst[i] = loop ? N : b;
      if (out_w) out_w[i] = attr ? attr[i * attr_stride] : 1.f;
    } else {
      out_src[i] = i - E;
      out_dst[i] = i - E;
      if (out_w) out_w[i] = 1.f;
    }
  }
}
}  // namespace mlgnn

extern "C" int mlgnn_sage_rewrite(const int64_t* edge_index, const float* edge_attr, int64_t attr_stride, int64_t E,
                                  int64_t N, int64_t* out_edge_index, float* out_weight, void* stream) {
  if (N < 0 || E < 0 || N > INT32_MAX - 2 || E + N > INT32_MAX || (edge_attr && attr_stride < 1)) return MLGNN_E_SHAPE;
  if (E + N == 0) return 0;
  if ((E > 0 && !edge_index) || !out_edge_index) return MLGNN_E_NULL;
  const int64_t n = E + N;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sage_rewrite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, edge_index,
                     edge_index + E, edge_attr, attr_stride, out_edge_index, out_edge_index + n, out_weight, E, N);
  return (int)hipGetLastError();
}

// B block-diagonal copies of one graph (the fold-constant gene network every sample of a TCGA batch carries,
// dataloader/multiloader.py:687-691): the CSR of the batch IS the CSR of the single graph with node ids shifted by b n and
// edge positions by b e -- the stable orderings are preserved copy by copy -- so it is written in one stream instead of
// being sorted out of the B-fold edge list again.
namespace mlgnn {
__global__ __launch_bounds__(256) void csr_replicate_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                           const int* __restrict__ eid, const int* __restrict__ rowptr_t,
                                                           const int* __restrict__ col_t, const int* __restrict__ pos_t,
                                                           const int* __restrict__ eid_t, int* __restrict__ o_rowptr,
                                                           int* __restrict__ o_col, int* __restrict__ o_eid,
                                                           int* __restrict__ o_rowptr_t, int* __restrict__ o_col_t,
                                                           int* __restrict__ o_pos_t, int* __restrict__ o_eid_t, int n, int e,
                                                           int copies) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t total_e = (int64_t)copies * e, total_n = (int64_t)copies * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_e + total_n + 1; i += stride) {
    if (i < total_e) {
      const int b = (int)(i / e), k = (int)(i - (int64_t)b * e);
      o_col[i] = col[k] + b * n;
      o_eid[i] = eid[k] + b * e;
      o_col_t[i] = col_t[k] + b * n;
      o_pos_t[i] = pos_t[k] + b * e;
      o_eid_t[i] = eid_t[k] + b * e;
    } else {
      const int64_t r = i - total_e;                      // 0 .. copies * n
      const int b = (int)(r / n), k = (int)(r - (int64_t)b * n);
      // (r = copies * n: b = copies, k = 0 -> rowptr[0] + copies * e = the total)
      o_rowptr[r] = rowptr[k] + b * e;
      o_rowptr_t[r] = rowptr_t[k] + b * e;
    }
  }
}
}  // namespace mlgnn

extern "C" int mlgnn_csr_replicate(const int32_t* rowptr, const int32_t* col, const int32_t* eid, const int32_t* rowptr_t,
                                   const int32_t* col_t, const int32_t* pos_t, const int32_t* eid_t, int32_t* out_rowptr,
                                   int32_t* out_col, int32_t* out_eid, int32_t* out_rowptr_t, int32_t* out_col_t,
                                   int32_t* out_pos_t, int32_t* out_eid_t, int64_t N, int64_t E, int64_t copies,
                                   void* stream) {
  if (N <= 0 || E < 0 || copies < 1 || N * copies > INT32_MAX - 1 || E * copies > INT32_MAX) return MLGNN_E_SHAPE;
  if (!rowptr || !rowptr_t || !out_rowptr || !out_rowptr_t) return MLGNN_E_NULL;
  if (E > 0 && (!col || !eid || !col_t || !pos_t || !eid_t || !out_col || !out_eid || !out_col_t || !out_pos_t || !out_eid_t))
    return MLGNN_E_NULL;
  const int64_t n = (E + N) * copies + 1;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(csr_replicate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, eid,
                     rowptr_t, col_t, pos_t, eid_t, out_rowptr, out_col, out_eid, out_rowptr_t, out_col_t, out_pos_t,
                     out_eid_t, (int)N, (int)E, (int)copies);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_edge_table_to_csr(const float* attr, int64_t row_stride, int64_t r, int64_t width,
                                       const int32_t* eid, const int32_t* eid_t, float* by_dst, float* by_src,
                                       int64_t E, void* stream) {
  if (E < 0 || r < 1 || width < r || width > 8 || row_stride < r) return MLGNN_E_SHAPE;
  if (E == 0) return 0;
  if (!attr || !eid || !eid_t || !by_dst || !by_src) return MLGNN_E_NULL;
  const int threads = 256;
  hipStream_t s = (hipStream_t)stream;
  int64_t done = 0;
  if (width == 1 && (((uintptr_t)eid | (uintptr_t)eid_t | (uintptr_t)by_dst | (uintptr_t)by_src) & 15) == 0 && E >= 4) {
    const int64_t n4 = E / 4;
    hipLaunchKernelGGL(edge_table_w1_kernel, dim3((unsigned)((n4 + threads - 1) / threads)), dim3(threads), 0, s, attr,
                       row_stride, (const int4*)eid, (const int4*)eid_t, (float4*)by_dst, (float4*)by_src, n4);
    done = n4 * 4;
  }
  if (done < E)
    hipLaunchKernelGGL(edge_table_kernel, dim3((unsigned)((E - done + threads - 1) / threads)), dim3(threads), 0, s, attr,
                       row_stride, (int)r, (int)width, eid + done, eid_t + done, by_dst + done * width,
                       by_src + done * width, E - done);
  return (int)hipGetLastError();
}
