// Backward of the MAX aggregator from compact winner lists.
//
// Reference: gcn_aggr='max' is the DEFAULT aggregator of the reference (opt.py:144; GenMessagePassing.aggregate,
// models/gcn_lib/sparse/torch_message.py:46-47 -> scatter(..., reduce='max')); its autograd sends the cotangent of
// (node i, channel c) to the ONE incoming edge that won the maximum there.  The general backward
// (csrc/aggregate_bwd.hip) walks the by-source CSR and gathers, for EVERY edge, the whole cotangent row and the whole
// winner row of its destination (640 bytes at d = 128) to keep the 1 / in-degree of the channels that edge won -- and the
// table-gradient pass of csrc/embedding.hip gathers the same rows once more in table-row order.  Here the winners are
// made compact first:
//
//   A  max_winners_kernel (streaming, by destination row): the d channels of row i are sorted by winning edge -- a
//      counting sort over the row's <= 256 incoming edges in LDS -- into wval[i][.] (cotangent values) and wch[i][.]
//      (their channels, one byte each); meta[p] = {offset, count} of edge p's run inside its row (p = by-destination
//      position).  Reads grad_out and argmax once (N d 8 bytes), writes N d 5 + E 4 bytes.
//   B  max_sparse_bwd_kernel (by source row j): grad_x[j][ch] += val over the runs of j's outgoing edges -- about
//      d / in-degree (value, channel) pairs per edge, ONE 64-byte line instead of ten.
//   C  max_sparse_table_grad_kernel (by table row t): the same sum over the edges that read table row t
//      (the edge-type embedding of global_edge='onehot', deepergcn.py:103-104).
//
// Eight lanes work on one edge (its run is 8 pairs on average at in-degree 16; longer runs take further rounds), eight
// rows (B) or eight interleaved shares of one table row (C) per wavefront, each with an accumulator row of its own in
// LDS: the lanes of one instruction never meet in one word (the channels of a run are distinct, rows do not share
// accumulators), every accumulator sees its edges in CSR order, shares are added in share order -- no atomics between
// workgroups, bitwise reproducible.  The forward names a winner only where its relu is active (argmax = -1 otherwise,
// csrc/aggregate_fwd.hip), so neither x nor the edge term is read again.
//
// Rows are short by contract: the caller takes this path only for graphs known to have no row longer than HUB_CAP
// (256) edges in either direction (mlgnn.CSRGraph.hub_tables); a longer destination row would not fit the LDS bins and is
// cut (its edges beyond 256 get empty runs) rather than read out of bounds.
#include "aggregate_short.h"
#include "mlgnn.h"

namespace mlgnn {

constexpr int kMsBins = 256;           // most incoming edges of one destination row
constexpr int kMsBlock = 256;
constexpr int kMsWaves = kMsBlock / kWave;

// LDS traffic between the lanes of ONE wavefront: DS instructions of a wave execute in order, the compiler is told not to
// move accesses across this point
__device__ __forceinline__ void ms_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- A: winners of a destination row, sorted by edge ---------------------------------------------------------------
// lane group of lpr = 2^lpr_log2 >= d / 4 lanes per row (8 .. 64), four channels per lane
__global__ __launch_bounds__(kMsBlock) void max_winners_kernel(const float* __restrict__ go, const int* __restrict__ argmax,
                                                              const int* __restrict__ rowptr, float* __restrict__ wval,
                                                              uint8_t* __restrict__ wch, uint32_t* __restrict__ meta,
                                                              int N, int d, int lpr_log2) {
  __shared__ uint32_t bins_all[kMsWaves][8][kMsBins];            // counts, then run starts (groups <= 8: d >= 32)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int lpr = 1 << lpr_log2, groups = kWave >> lpr_log2;
  const int sub = lane >> lpr_log2, cl = lane & (lpr - 1);
  uint32_t* bins = bins_all[wave][sub];
  const int per = kMsBins >> lpr_log2;                           // bins a lane scans: 4 .. 32
  const bool cact = 4 * cl < d;
  const int rows_per_block = kMsWaves * groups;
  int first, last;
  short_rows_of_block(N, rows_per_block, first, last);
  const int r = first + wave * groups + sub;
  if (first >= last) return;                                     // (whole workgroup)
  const bool ract = r < last;
  const int beg = ract ? rowptr[r] : 0;
  const int deg = ract ? min(rowptr[r + 1] - beg, kMsBins) : 0;
  for (int b = cl; b < deg; b += lpr) bins[b] = 0u;
  int am[4] = {-1, -1, -1, -1};
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  if (ract && cact) {
    const size_t at = (size_t)r * d + 4 * cl;
    load_vec<4>(am, argmax + at);
    load_vec<4>(g, go + at);
  }
  ms_wave_sync();
  int slot[4], rank[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    slot[i] = am[i] >= 0 ? am[i] - beg : -1;
    if (slot[i] >= deg) slot[i] = -1;                            // (a cut row: see the header)
    rank[i] = slot[i] >= 0 ? (int)atomicAdd(&bins[slot[i]], 1u) : 0;
  }
  ms_wave_sync();
  // exclusive scan of the counts: a lane owns `per` consecutive bins
  uint32_t total = 0;
  const int b0 = cl * per;
  for (int j = 0; j < per; ++j) total += (b0 + j < deg) ? bins[b0 + j] : 0u;
  uint32_t incl = total;
  for (int off = 1; off < lpr; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, lpr);
    if (cl >= off) incl += up;
  }
  uint32_t run = incl - total;
  for (int j = 0; j < per; ++j) {
    if (b0 + j < deg) {
      const uint32_t c = bins[b0 + j];
      bins[b0 + j] = run;
      meta[beg + b0 + j] = run | (c << 16);
      run += c;
    }
  }
  if (ract) {
    // (edges of a cut row beyond the bins: empty runs)
    const int full = rowptr[r + 1] - beg;
    for (int b = kMsBins + cl; b < full; b += lpr) meta[beg + b] = 0u;
  }
  ms_wave_sync();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (slot[i] >= 0) {
      const size_t at = (size_t)r * d + bins[slot[i]] + rank[i];
      wval[at] = g[i];
      wch[at] = (uint8_t)(4 * cl + i);
    }
  }
}

// One chunk of eight edges of one share: the lanes hold (destination row, meta) of edge k = lane & 7 each; the runs of
// the eight edges are added to `acc` edge after edge.  Loads of all eight edges are issued before the first use.
__device__ __forceinline__ void ms_add_chunk(const float* __restrict__ wval, const uint8_t* __restrict__ wch, int dst, uint32_t m,
                                             int d, float* acc) {
  const int k = threadIdx.x & 7;
  uint32_t longest = m >> 16;
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, off, 8));
  // (rounds are uniform over the wave: every share runs as many as the longest run of its chunk needs; a share past its
  // own needs loads nothing)
  uint32_t wave_longest = longest;
#pragma unroll
  for (int off = 8; off < kWave; off <<= 1) wave_longest = max(wave_longest, (uint32_t)__shfl_xor((int)wave_longest, off));
  for (uint32_t base = 0; base < wave_longest; base += 8) {
    float v[8];
    int ch[8];
    bool on[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int dj = __shfl(dst, j, 8);
      const uint32_t mj = (uint32_t)__shfl((int)m, j, 8);
      const uint32_t cnt = mj >> 16, kk = base + (uint32_t)k;
      on[j] = kk < cnt;
      v[j] = 0.f; ch[j] = 0;
      if (on[j]) {
        const size_t at = (size_t)dj * d + (mj & 0xffffu) + kk;
        v[j] = wval[at];
        ch[j] = wch[at];
      }
    }
    // one ds_add_f32 per edge: its lanes hold distinct channels of ONE edge per share (no two lanes of an instruction meet
    // in a word), and the DS unit executes a wave's instructions in order (the next edge may name the same channels)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (on[j]) __hip_atomic_fetch_add(acc + ch[j], v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  ms_wave_sync();
}

// ---- B: grad_x by source row ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kMsBlock) void max_sparse_bwd_kernel(const float* __restrict__ wval, const uint8_t* __restrict__ wch,
                                                                 const uint32_t* __restrict__ meta,
                                                                 const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
                                                                 const int* __restrict__ pos_t, const float* __restrict__ root,
                                                                 float* __restrict__ gx, int N, int d) {
  extern __shared__ float ms_acc[];                              // [kMsWaves][8][d]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int s = lane >> 3, k = lane & 7;
  float* acc = ms_acc + (size_t)(wave * 8 + s) * d;
  int first, last;
  short_rows_of_block(N, kMsWaves * 8, first, last);
  if (first >= last) return;
  const int r = first + wave * 8 + s;
  const bool ract = r < last;
  for (int c = 4 * k; c < d; c += 32) *reinterpret_cast<float4*>(acc + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  ms_wave_sync();
  const int beg = ract ? rowptr_t[r] : 0, end = ract ? rowptr_t[r + 1] : 0;
  int longest = end - beg;
#pragma unroll
  for (int off = 8; off < kWave; off <<= 1) longest = max(longest, __shfl_xor(longest, off));
  for (int base = 0; base < longest; base += 8) {                // (uniform over the wave)
    const int q = beg + base + k;
    int dst = 0;
    uint32_t m = 0u;
    if (q < end) {
      dst = col_t[q];
      m = meta[pos_t[q]];
    }
    ms_add_chunk(wval, wch, dst, m, d, acc);
  }
  if (ract) {
    for (int c = 4 * k; c < d; c += 32) {
      float4 v = *reinterpret_cast<const float4*>(acc + c);
      if (root) {                                                // GENConv's h = x + m from the same aggregation call
        const float4 o = *reinterpret_cast<const float4*>(root + (size_t)r * d + c);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<float4*>(gx + (size_t)r * d + c) = v;
    }
  }
}

// ---- C: gradient of a table edge term, one wavefront per table row ----------------------------------------------------
// edges: by-destination positions sorted (stably) by table row (pos_s), their destination rows (dst_s); the eight shares
// of the wave take the chunks of eight edges round robin and are added in share order at the end
__global__ __launch_bounds__(kMsBlock) void max_sparse_table_grad_kernel(const float* __restrict__ wval, const uint8_t* __restrict__ wch,
                                                                        const uint32_t* __restrict__ meta,
                                                                        const int* __restrict__ dst_s, const int* __restrict__ pos_s,
                                                                        const int* __restrict__ rowptr, float* __restrict__ out,
                                                                        int T, int d, int accumulate) {
  extern __shared__ float ms_acc[];                              // [kMsWaves][8][d]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int s = lane >> 3, k = lane & 7;
  float* wave_acc = ms_acc + (size_t)wave * 8 * d;
  float* acc = wave_acc + (size_t)s * d;
  const int n_waves = gridDim.x * kMsWaves;
  for (int t = blockIdx.x * kMsWaves + wave; t < T; t += n_waves) {
    for (int c = 4 * k; c < d; c += 32) *reinterpret_cast<float4*>(acc + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    ms_wave_sync();
    const int beg = rowptr[t], end = rowptr[t + 1];
    for (int base = beg; base < end; base += 64) {               // (uniform over the wave)
      const int q = base + 8 * s + k;
      int dst = 0;
      uint32_t m = 0u;
      if (q < end) {
        dst = dst_s[q];
        m = meta[pos_s[q]];
      }
      ms_add_chunk(wval, wch, dst, m, d, acc);
    }
    // shares in order
    for (int c = lane; c < d; c += kWave) {
      float sum = wave_acc[c];
#pragma unroll
      for (int j = 1; j < 8; ++j) sum += wave_acc[(size_t)j * d + c];
      float* o = out + (size_t)t * d + c;
      *o = accumulate ? *o + sum : sum;
    }
    ms_wave_sync();
  }
}

static bool ms_shape_ok(int64_t N, int64_t d) {
  return N > 0 && N <= INT32_MAX && d >= 32 && d <= 256 && d % 4 == 0 && N * d < ((int64_t)1 << 40);
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_max_sparse_supported(int64_t N, int64_t d) { return ms_shape_ok(N, d) ? 1 : 0; }

extern "C" int mlgnn_max_winners(const float* grad_out, const int32_t* argmax, const int32_t* rowptr, float* wval, void* wch,
                                 void* meta, int64_t N, int64_t d, void* stream) {
  if (!ms_shape_ok(N, d)) return MLGNN_E_SHAPE;
  if (!grad_out || !argmax || !rowptr || !wval || !wch || !meta) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(argmax)) & 15) != 0) return MLGNN_E_ALIGN;
  const int lpr_log2 = lanes_per_row_log2(d, 4);
  const int groups = kWave >> lpr_log2;
  hipLaunchKernelGGL(max_winners_kernel, dim3((unsigned)short_grid(N, kMsWaves * groups)), dim3(kMsBlock), 0, (hipStream_t)stream,
                     grad_out, argmax, rowptr, wval, static_cast<uint8_t*>(wch), static_cast<uint32_t*>(meta), (int)N, (int)d,
                     lpr_log2);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_max_sparse_bwd(const float* wval, const void* wch, const void* meta, const int32_t* rowptr_t,
                                    const int32_t* col_t, const int32_t* pos_t, const float* root, float* grad_x, int64_t N,
                                    int64_t d, void* stream) {
  if (!ms_shape_ok(N, d)) return MLGNN_E_SHAPE;
  if (!wval || !wch || !meta || !rowptr_t || !col_t || !pos_t || !grad_x) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(root)) & 15) != 0) return MLGNN_E_ALIGN;
  const size_t lds = (size_t)kMsWaves * 8 * d * sizeof(float);
  hipLaunchKernelGGL(max_sparse_bwd_kernel, dim3((unsigned)short_grid(N, kMsWaves * 8)), dim3(kMsBlock), lds, (hipStream_t)stream,
                     wval, static_cast<const uint8_t*>(wch), static_cast<const uint32_t*>(meta), rowptr_t, col_t, pos_t, root,
                     grad_x, (int)N, (int)d);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_max_sparse_table_grad(const float* wval, const void* wch, const void* meta, const int32_t* dst_sorted,
                                           const int32_t* pos_sorted, const int32_t* rowptr, float* grad_table, int64_t N,
                                           int64_t d, int64_t T, int accumulate, void* stream) {
  if (!ms_shape_ok(N, d) || T < 0 || T > INT32_MAX) return MLGNN_E_SHAPE;
  if (T == 0) return 0;
  if (!wval || !wch || !meta || !dst_sorted || !pos_sorted || !rowptr || !grad_table) return MLGNN_E_NULL;
  const size_t lds = (size_t)kMsWaves * 8 * d * sizeof(float);
  int64_t blocks = (T + kMsWaves - 1) / kMsWaves;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  hipLaunchKernelGGL(max_sparse_table_grad_kernel, dim3((unsigned)blocks), dim3(kMsBlock), lds, (hipStream_t)stream, wval,
                     static_cast<const uint8_t*>(wch), static_cast<const uint32_t*>(meta), dst_sorted, pos_sorted, rowptr,
                     grad_table, (int)T, (int)d, accumulate);
  return (int)hipGetLastError();
}
