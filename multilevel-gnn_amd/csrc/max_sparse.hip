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
//      counting sort over the row's <= 256 incoming edges in LDS -- into 8-byte records {cotangent value, channel};
//      meta[p] = {first record, count} of edge p's run (p = by-destination position).  Row i owns the records from
//      even(i (d + 2) + rowptr[i]) on and every run starts on an even record (odd runs are padded by one), so a lane reads
//      TWO records with one 16-byte load.  Reads grad_out and argmax once (N d 8 bytes), writes N d 8 + E 8 bytes.
//   B  max_sparse_bwd_kernel (by source row j): grad_x[j][ch] += val over the runs of j's outgoing edges -- about
//      d / in-degree (value, channel) pairs per edge, one or two 64-byte lines instead of ten, and ONE vector-memory
//      instruction per eight edges (the kernel is bound by the number of those, not by bytes).
//   C  max_sparse_table_grad_kernel (by table row t): the same sum over the edges that read table row t
//      (the edge-type embedding of global_edge='onehot', deepergcn.py:103-104).
//
// Eight lanes work on one edge (two records each: runs of up to 16 pairs in one round -- 8 on average at in-degree 16;
// longer runs take further rounds), eight
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
constexpr int kMsMaxBlocks = 256 * 8;  // persistent workgroups: eight per CU

// LDS traffic between the lanes of ONE wavefront: DS instructions of a wave execute in order, the compiler is told not to
// move accesses across this point
__device__ __forceinline__ void ms_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// units [u, u_end) step `stride` of this workgroup: XCD x owns the x-th eighth of the units (gridDim.x is a multiple of 8)
__device__ __forceinline__ void ms_units_of_block(int n_units, int& u, int& u_end, int& stride) {
  const int xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds;
  const int per_xcd = (n_units + kXcds - 1) / kXcds;
  const int lo = xcd * per_xcd;
  u = lo + slot;
  u_end = min(n_units, lo + per_xcd);
  stride = gridDim.x / kXcds;
}

// persistent grid: every workgroup resident at once (a second, partly filled round of workgroups would add its whole
// length to the launch), a multiple of 8; `slot` caches the occupancy of one kernel at one LDS size
struct MsOcc { size_t lds = ~(size_t)0; int per_cu = 0; int cus = 0; };
template <typename K>
static int ms_grid(K kernel, size_t lds, int64_t n_units, MsOcc& slot) {
  if (slot.lds != lds) {
    int n = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kMsBlock, lds) != hipSuccess || n < 1) n = 1;
    slot.cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    slot.per_cu = n;
    slot.lds = lds;
  }
  int64_t b = (n_units + kXcds - 1) / kXcds * kXcds;
  const int64_t cap = (int64_t)slot.cus * slot.per_cu / kXcds * kXcds;
  if (b > cap) b = cap;
  if (b > kMsMaxBlocks) b = kMsMaxBlocks;
  if (b < kXcds) b = kXcds;
  return (int)b;
}

// first record of destination row r (even)
__device__ __forceinline__ uint32_t ms_row_base(int r, int d, int beg) {
  return ((uint32_t)r * (uint32_t)(d + 2) + (uint32_t)beg + 1u) & ~1u;
}

// ---- A: winners of a destination row, sorted by edge ---------------------------------------------------------------
// lane group of lpr = 2^lpr_log2 >= d / 4 lanes per row (8 .. 64), four channels per lane
__global__ __launch_bounds__(kMsBlock) void max_winners_kernel(const float* __restrict__ go, const int* __restrict__ argmax,
                                                              const int* __restrict__ rowptr, uint2* __restrict__ wrec,
                                                              uint2* __restrict__ meta, int N, int d, int lpr_log2) {
  extern __shared__ uint32_t ms_bins[];                          // [kMsWaves][groups][kMsBins]: counts, then run starts
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int lpr = 1 << lpr_log2, groups = kWave >> lpr_log2;
  const int sub = lane >> lpr_log2, cl = lane & (lpr - 1);
  uint32_t* bins = ms_bins + (size_t)(wave * groups + sub) * kMsBins;
  const bool cact = 4 * cl < d;
  const int rows_per_unit = kMsWaves * groups;
  int u, u_end, stride;
  ms_units_of_block((N + rows_per_unit - 1) / rows_per_unit, u, u_end, stride);
  // the operands of the NEXT row are requested before this row's LDS phases (a row is two dependent round trips --
  // row pointer, then cotangent and argmax -- and a few hundred cycles of LDS work: un-pipelined the kernel waits)
  int n_beg = 0, n_end = 0;                                      // (subtracted at the use: nothing here waits for a load)
  int n_am[4] = {-1, -1, -1, -1};
  float n_g[4] = {0.f, 0.f, 0.f, 0.f};
  auto request = [&](int unit) {
    const int r = unit * rows_per_unit + wave * groups + sub;
    n_beg = 0; n_end = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { n_am[i] = -1; n_g[i] = 0.f; }
    if (unit < u_end && r < N) {
      n_beg = rowptr[r];
      n_end = rowptr[r + 1];
      if (cact) {
        const size_t at = (size_t)r * d + 4 * cl;
        load_vec<4>(n_am, argmax + at);
        load_vec<4>(n_g, go + at);
      }
    }
  };
  request(u);
  for (; u < u_end; u += stride) {
    const int r = u * rows_per_unit + wave * groups + sub;
    const int beg = n_beg, full = n_end - n_beg;
    int am[4];
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { am[i] = n_am[i]; g[i] = n_g[i]; }
    request(u + stride);
    const int deg = min(full, kMsBins);
    const uint32_t base = ms_row_base(r, d, beg);
    for (int b = cl; b < deg; b += lpr) bins[b] = 0u;
    ms_wave_sync();
    int slot[4], rank[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      slot[i] = am[i] >= 0 ? am[i] - beg : -1;
      if (slot[i] >= deg) slot[i] = -1;                          // (a cut row: see the header)
      rank[i] = slot[i] >= 0 ? (int)atomicAdd(&bins[slot[i]], 1u) : 0;
    }
    ms_wave_sync();
    // exclusive scan of the (even-padded) counts, lpr bins per round -- ONE round for a row of up to lpr edges (32 at
    // d = 128); a lane's bin is its edge: the {first record, count} stores of a round are one coalesced instruction
    uint32_t carry = 0;
    for (int b0 = 0; b0 < deg; b0 += lpr) {
      const int b = b0 + cl;
      const uint32_t c = b < deg ? bins[b] : 0u;
      const uint32_t pc = (c + 1u) & ~1u;
      uint32_t incl = pc;
      for (int off = 1; off < lpr; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, lpr);
        if (cl >= off) incl += up;
      }
      const uint32_t start = carry + incl - pc;
      if (b < deg) {
        bins[b] = start;
        meta[beg + b] = make_uint2(base + start, c);
      }
      carry += (uint32_t)__shfl((int)incl, lpr - 1, lpr);
    }
    for (int b = kMsBins + cl; b < full; b += lpr) meta[beg + b] = make_uint2(0u, 0u);     // (a cut row: empty runs)
    ms_wave_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (slot[i] >= 0) wrec[base + bins[slot[i]] + rank[i]] = make_uint2(__builtin_bit_cast(uint32_t, g[i]), (uint32_t)(4 * cl + i));
    ms_wave_sync();
  }
}

// One chunk of eight edges of one share: lane k = lane & 7 holds {first record, count} of edge k; the runs of the eight
// edges are added to `acc` edge after edge, two records per lane and round.  The loads of all eight edges are issued
// before the first use.
__device__ __forceinline__ void ms_add_chunk(const uint2* __restrict__ wrec, uint2 m, float* acc) {
  const uint32_t k2 = 2u * (uint32_t)(threadIdx.x & 7);
  uint32_t longest = m.y;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, off));
  // (rounds are uniform over the wave: as many as its longest run needs -- one, with rare exceptions)
  for (uint32_t base = 0; base < longest; base += 16) {
    uint4 rec[8];
    uint32_t left[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t oj = (uint32_t)__shfl((int)m.x, j, 8), cj = (uint32_t)__shfl((int)m.y, j, 8);
      const uint32_t kk = base + k2;
      left[j] = cj > kk ? cj - kk : 0u;                          // records of this lane's pair that exist: 0, 1, >= 2
      rec[j] = make_uint4(0u, 0u, 0u, 0u);
      if (left[j]) rec[j] = *reinterpret_cast<const uint4*>(wrec + oj + kk);       // (oj and kk even: 16-byte aligned)
    }
    // per edge: the lanes of a share hold distinct channels of ONE edge (no two lanes of an instruction meet in a word) and
    // the DS unit executes a wave's instructions in order (the next edge may name the same channels): plain read - add -
    // write, no LDS atomics (ds_add_f32 kept the LDS index unit busy for 68 % of the launch: tools/pmc_max_sparse.sh)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a0 = 0.f, a1 = 0.f;
      if (left[j] >= 1u) a0 = acc[rec[j].y];
      if (left[j] >= 2u) a1 = acc[rec[j].w];
      if (left[j] >= 1u) acc[rec[j].y] = a0 + __builtin_bit_cast(float, rec[j].x);
      if (left[j] >= 2u) acc[rec[j].w] = a1 + __builtin_bit_cast(float, rec[j].z);
      ms_wave_sync();
    }
  }
  ms_wave_sync();
}

// ---- B: grad_x by source row ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kMsBlock) void max_sparse_bwd_kernel(const uint2* __restrict__ wrec, const uint2* __restrict__ meta,
                                                                 const int* __restrict__ rowptr_t, const int* __restrict__ pos_t,
                                                                 const float* __restrict__ root, float* __restrict__ gx, int N,
                                                                 int d) {
  extern __shared__ float ms_acc[];                              // [kMsWaves][8][d]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int s = lane >> 3, k = lane & 7;
  float* acc = ms_acc + (size_t)(wave * 8 + s) * d;
  constexpr int kRows = kMsWaves * 8;
  int u, u_end, stride;
  ms_units_of_block((N + kRows - 1) / kRows, u, u_end, stride);
  for (; u < u_end; u += stride) {
    const int r = u * kRows + wave * 8 + s;
    const bool ract = r < N;
    for (int c = 4 * k; c < d; c += 32) *reinterpret_cast<float4*>(acc + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    ms_wave_sync();
    const int beg = ract ? rowptr_t[r] : 0, end = ract ? rowptr_t[r + 1] : 0;
    int longest = end - beg;
#pragma unroll
    for (int off = 8; off < kWave; off <<= 1) longest = max(longest, __shfl_xor(longest, off));
    for (int base = 0; base < longest; base += 32) {             // (uniform over the wave)
      // {first record, count} of up to 32 edges per row first (four independent gathers in flight), then chunk by chunk
      uint2 m[4];
      int p[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int q = beg + base + 8 * c + k;
        p[c] = q < end ? pos_t[q] : -1;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = p[c] >= 0 ? meta[p[c]] : make_uint2(0u, 0u);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (base + 8 * c < longest) ms_add_chunk(wrec, m[c], acc);
    }
    if (ract) {
      for (int c = 4 * k; c < d; c += 32) {
        float4 v = *reinterpret_cast<const float4*>(acc + c);
        if (root) {                                              // GENConv's h = x + m from the same aggregation call
          const float4 o = *reinterpret_cast<const float4*>(root + (size_t)r * d + c);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(gx + (size_t)r * d + c) = v;
      }
    }
    ms_wave_sync();
  }
}

// ---- C: gradient of a table edge term, one wavefront per table row ----------------------------------------------------
// edges: by-destination positions sorted (stably) by table row (pos_s); the eight shares of the wave take the chunks of
// eight edges round robin and are added in share order at the end
__global__ __launch_bounds__(kMsBlock) void max_sparse_table_grad_kernel(const uint2* __restrict__ wrec, const uint2* __restrict__ meta,
                                                                        const int* __restrict__ pos_s, const int* __restrict__ rowptr,
                                                                        float* __restrict__ out, int T, int d, int accumulate) {
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
    for (int base = beg; base < end; base += 256) {              // (uniform over the wave)
      uint2 m[4];
      int p[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int q = base + 64 * c + 8 * s + k;
        p[c] = q < end ? pos_s[q] : -1;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = p[c] >= 0 ? meta[p[c]] : make_uint2(0u, 0u);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (base + 64 * c < end) ms_add_chunk(wrec, m[c], acc);
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
  return N > 0 && N <= INT32_MAX && d >= 32 && d <= 256 && d % 4 == 0 && N * (d + 2) < ((int64_t)1 << 31);
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_max_sparse_supported(int64_t N, int64_t d) { return ms_shape_ok(N, d) ? 1 : 0; }

extern "C" int64_t mlgnn_max_sparse_records(int64_t N, int64_t d, int64_t E) {
  if (!ms_shape_ok(N, d) || E < 0 || N * (d + 2) + E + 2 >= ((int64_t)1 << 32)) return MLGNN_E_SHAPE;
  return N * (d + 2) + E + 2;
}

extern "C" int mlgnn_max_winners(const float* grad_out, const int32_t* argmax, const int32_t* rowptr, void* records, void* meta,
                                 int64_t N, int64_t d, void* stream) {
  if (!ms_shape_ok(N, d)) return MLGNN_E_SHAPE;
  if (!grad_out || !argmax || !rowptr || !records || !meta) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(argmax) | reinterpret_cast<uintptr_t>(records)) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(meta) & 7) != 0)
    return MLGNN_E_ALIGN;
  const int lpr_log2 = lanes_per_row_log2(d, 4);
  const int rows_per_unit = kMsWaves * (kWave >> lpr_log2);
  const size_t lds = (size_t)rows_per_unit * kMsBins * sizeof(uint32_t);
  static MsOcc occ;
  hipLaunchKernelGGL(max_winners_kernel, dim3((unsigned)ms_grid(max_winners_kernel, lds, (N + rows_per_unit - 1) / rows_per_unit, occ)),
                     dim3(kMsBlock), lds, (hipStream_t)stream, grad_out, argmax, rowptr, static_cast<uint2*>(records), static_cast<uint2*>(meta), (int)N,
                     (int)d, lpr_log2);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_max_sparse_bwd(const void* records, const void* meta, const int32_t* rowptr_t, const int32_t* pos_t,
                                    const float* root, float* grad_x, int64_t N, int64_t d, void* stream) {
  if (!ms_shape_ok(N, d)) return MLGNN_E_SHAPE;
  if (!records || !meta || !rowptr_t || !pos_t || !grad_x) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(root) | reinterpret_cast<uintptr_t>(records)) & 15) != 0)
    return MLGNN_E_ALIGN;
  const size_t lds = (size_t)kMsWaves * 8 * d * sizeof(float);
  static MsOcc occ;
  hipLaunchKernelGGL(max_sparse_bwd_kernel,
                     dim3((unsigned)ms_grid(max_sparse_bwd_kernel, lds, (N + kMsWaves * 8 - 1) / (kMsWaves * 8), occ)), dim3(kMsBlock), lds,
                     (hipStream_t)stream, static_cast<const uint2*>(records), static_cast<const uint2*>(meta), rowptr_t, pos_t, root,
                     grad_x, (int)N, (int)d);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_max_sparse_table_grad(const void* records, const void* meta, const int32_t* pos_sorted, const int32_t* rowptr,
                                           float* grad_table, int64_t N, int64_t d, int64_t T, int accumulate, void* stream) {
  if (!ms_shape_ok(N, d) || T < 0 || T > INT32_MAX) return MLGNN_E_SHAPE;
  if (T == 0) return 0;
  if (!records || !meta || !pos_sorted || !rowptr || !grad_table) return MLGNN_E_NULL;
  if ((reinterpret_cast<uintptr_t>(records) & 15) != 0) return MLGNN_E_ALIGN;
  const size_t lds = (size_t)kMsWaves * 8 * d * sizeof(float);
  static MsOcc occ;
  const int blocks = ms_grid(max_sparse_table_grad_kernel, lds, (T + kMsWaves - 1) / kMsWaves, occ);
  hipLaunchKernelGGL(max_sparse_table_grad_kernel, dim3((unsigned)blocks), dim3(kMsBlock), lds, (hipStream_t)stream,
                     static_cast<const uint2*>(records), static_cast<const uint2*>(meta), pos_sorted, rowptr, grad_table, (int)T,
                     (int)d, accumulate);
  return (int)hipGetLastError();
}
