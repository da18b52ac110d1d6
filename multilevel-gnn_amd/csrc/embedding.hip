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


// ---- table gradient of the max aggregator from the DESTINATION side -------------------------------------------------
// Under max only the winning edge of (i, c) carries gradient, and the forward names it: argmax[i][c] = its position in
// the by-destination edge order, or -1 where no gradient flows (no incoming edge, or the winner sits on relu's flat
// side).  So  grad_table[t][c] = sum over { i : rows_by_dst[argmax[i][c]] = t } of grad_out[i][c]  -- a STREAMING pass
// over two [N, d] arrays (the winners of one destination row are edges of that row: their table rows sit in one or two
// cache lines) instead of an [E, d] per-edge gradient written, re-read per layer and reduced (5.2 GB per layer for a
// BASELINE configs[1] batch).  A thread owns one column quad and walks rows; its sums live in LDS slots of its own
// ([T][4][256] floats: no atomics, no conflicts), the threads of a workgroup that share a column quad are added in thread
// order, workgroups leave partial tables that reduce_partials adds in workgroup order: bitwise reproducible.
using emb_f4 = __attribute__((ext_vector_type(4))) float;

struct MaxTableArgs {
  const float* go; const int* argmax; const int* rows_dst; float* partials;
  int N; int d; int T; int rows_per_block;
};

__global__ __launch_bounds__(256) void max_table_grad_kernel(const MaxTableArgs a) {
  extern __shared__ float tab_acc[];                       // [T][4][256]
  const int tid = threadIdx.x;
  const int Q = a.d >> 2;                                  // column quads per row (<= 256)
  const int rpp = 256 / Q;                                 // rows per pass of the workgroup
  for (int k = tid; k < a.T * 1024; k += 256) tab_acc[k] = 0.f;
  __syncthreads();
  const int rr = tid / Q, cq = tid - rr * Q;
  const int r_beg = blockIdx.x * a.rows_per_block, r_end = min(a.N, r_beg + a.rows_per_block);
  if (rr < rpp) {
    constexpr int kU = 4;
    for (int r = r_beg + rr; r < r_end; r += rpp * kU) {
      int4 am[kU];
      float4 g[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int ru = r + u * rpp;
        am[u] = make_int4(-1, -1, -1, -1);
        g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ru < r_end) {
          const size_t q = (size_t)ru * Q + cq;
          am[u] = reinterpret_cast<const int4*>(a.argmax)[q];
          const emb_f4 gv = __builtin_nontemporal_load(reinterpret_cast<const emb_f4*>(a.go) + q);   // read once
          g[u] = make_float4(gv.x, gv.y, gv.z, gv.w);
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int e[4] = {am[u].x, am[u].y, am[u].z, am[u].w};
        const float v[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
        int t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = e[i] >= 0 ? a.rows_dst[e[i]] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (t[i] >= 0) tab_acc[(t[i] * 4 + i) * 256 + tid] += v[i];
      }
    }
  }
  __syncthreads();
  // partial table of this workgroup: the rpp threads of a column quad in thread order
  float* part = a.partials + (size_t)blockIdx.x * a.T * a.d;
  for (int k = tid; k < a.T * a.d; k += 256) {
    const int t = k / a.d, c = k - t * a.d;
    const float* slot = tab_acc + (t * 4 + (c & 3)) * 256 + (c >> 2);
    float sum = 0.f;
    for (int j = 0; j < rpp; ++j) sum += slot[j * Q];
    part[k] = sum;
  }
}

constexpr int kMaxTableBlocks = 1024;

// ---- the same gradient for a LARGE table: one wavefront per table row --------------------------------------------------
// nn.Embedding(pathway_edge_num, hidden) has one row per (gene, pathway) membership of KEGG (multiloader.py:105-106,
// 991-1005: tens of thousands of rows, a handful of edges per row and graph), so per-workgroup partial tables do not fit
// anywhere.  The edges -- by-destination positions -- are sorted by table row once per batch (stable: destination order
// inside a row, so the wavefronts sweep the graphs of the batch together and the cotangent rows they gather stay in L2 /
// Infinity Cache); a wavefront walks the edges of its row, gathers the cotangent and argmax rows of each edge's
// destination and keeps the channels that edge won:  acc[c] += (argmax[i][c] == pos) ? grad_out[i][c] : 0.
// Nothing per edge is ever written (the [E, d] gradient this replaces: 5.2 GB written, re-read and reduced per layer at
// BASELINE configs[1] size).  Fixed order: bitwise reproducible.
// The winner of (i, c) is read as the ONE-BYTE slot the aggregation backward derived from argmax for its own gathers
// (max_slot_kernel, csrc/aggregate_bwd.hip: the winner's position inside row i) when that call left them valid
// (*spread == 0: no row longer than 254 edges): 640 instead of 1024 gathered bytes per edge.
struct MaxTypeArgs {
  const float* go; const int* argmax; const int* dst_s; const int* pos_s; const int* rel_s; const int* rowptr; float* out;
  const uint8_t* slot8; const int* spread;
  int T; int d; int lpr_log2; int accumulate;
};

template <bool SLOTS>
__device__ __forceinline__ void max_table_grad_by_type_body(const MaxTypeArgs& a) {
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
        int am[kUnroll][4], pos[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int kk = k + u * groups;
          pos[u] = -2;
#pragma unroll
          for (int i = 0; i < 4; ++i) { v[u][i] = 0.f; am[u][i] = -1; }
          if (kk < end) {
            const size_t row = (size_t)a.dst_s[kk] * a.d + c0;
            pos[u] = SLOTS ? a.rel_s[kk] : a.pos_s[kk];
            load_vec<4>(v[u], a.go + row);
            if constexpr (SLOTS) {
              const uint32_t w = *reinterpret_cast<const uint32_t*>(a.slot8 + row);
#pragma unroll
              for (int i = 0; i < 4; ++i) am[u][i] = (int)((w >> (8 * i)) & 0xffu);
            } else {
              load_vec<4>(am[u], a.argmax + row);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += (am[u][i] == pos[u]) ? v[u][i] : 0.f;
      }
      for (int off = lpr; off < kWave; off <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off);
      if (sub == 0 && cact) {
        float* o = a.out + (size_t)t * a.d + c0;
        if (a.accumulate) {
          float prev[4];
          load_vec<4>(prev, o);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += prev[i];
        }
        store_vec<4>(o, acc);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void max_table_grad_by_type_kernel(const MaxTypeArgs a) {
  if (a.slot8 != nullptr && *a.spread == 0) max_table_grad_by_type_body<true>(a);     // (uniform over the launch)
  else max_table_grad_by_type_body<false>(a);
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

static int max_table_blocks(int64_t N, int64_t d, int* rows_per_block) {
  const int rpp = 256 / (int)(d / 4);
  int64_t rpb = (N + kMaxTableBlocks - 1) / kMaxTableBlocks;
  rpb = (rpb + rpp - 1) / rpp * rpp;
  if (rpb < 4 * rpp) rpb = 4 * rpp;
  *rows_per_block = (int)rpb;
  return (int)((N + rpb - 1) / rpb);
}

extern "C" int mlgnn_max_table_grad_supported(int64_t N, int64_t d, int64_t T) {
  return N > 0 && N <= INT32_MAX && d >= 4 && d % 4 == 0 && d <= 1024 && N * d < ((int64_t)1 << 40) && T >= 1 && T <= 36;
}

extern "C" int64_t mlgnn_max_table_grad_workspace_floats(int64_t N, int64_t d, int64_t T) {
  if (!mlgnn_max_table_grad_supported(N, d, T)) return MLGNN_E_SHAPE;
  int rpb = 0;
  return (int64_t)max_table_blocks(N, d, &rpb) * T * d;
}

extern "C" int mlgnn_max_table_grad(const float* grad_out, const int32_t* argmax, const int32_t* rows_by_dst,
                                    float* grad_table, float* workspace, int64_t workspace_floats, int64_t N, int64_t d,
                                    int64_t T, int accumulate, void* stream) {
  if (!mlgnn_max_table_grad_supported(N, d, T)) return MLGNN_E_SHAPE;
  if (!grad_out || !argmax || !rows_by_dst || !grad_table || !workspace) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(argmax) | reinterpret_cast<uintptr_t>(workspace) |
        reinterpret_cast<uintptr_t>(grad_table)) & 15) != 0)
    return MLGNN_E_ALIGN;
  MaxTableArgs a;
  a.go = grad_out; a.argmax = argmax; a.rows_dst = rows_by_dst; a.partials = workspace;
  a.N = (int)N; a.d = (int)d; a.T = (int)T;
  const int blocks = max_table_blocks(N, d, &a.rows_per_block);
  if (workspace_floats < (int64_t)blocks * T * d) return MLGNN_E_WORKSPACE;
  const size_t lds = (size_t)T * 1024 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(max_table_grad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024 * (int)sizeof(float));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(max_table_grad_kernel, dim3((unsigned)blocks), dim3(256), lds, s, a);
  int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_table, blocks, (int)(T * d), s, accumulate != 0);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_max_table_grad_by_type(const float* grad_out, const int32_t* argmax, const int32_t* dst_sorted,
                                            const int32_t* pos_sorted, const int32_t* rel_sorted, const int32_t* rowptr,
                                            const void* slots, float* grad_table, int64_t N, int64_t d, int64_t T,
                                            int accumulate, void* stream) {
  if (N <= 0 || N > INT32_MAX || T < 0 || T > INT32_MAX || d <= 0 || d % 4 != 0 || d > 4096) return MLGNN_E_SHAPE;
  if (T == 0) return 0;
  if (!grad_out || !argmax || !dst_sorted || !pos_sorted || !rowptr || !grad_table) return MLGNN_E_NULL;
  if (((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(argmax) | reinterpret_cast<uintptr_t>(grad_table)) &
       15) != 0)
    return MLGNN_E_ALIGN;
  MaxTypeArgs a;
  a.go = grad_out; a.argmax = argmax; a.dst_s = dst_sorted; a.pos_s = pos_sorted; a.rowptr = rowptr; a.out = grad_table;
  a.rel_s = rel_sorted; a.slot8 = nullptr; a.spread = nullptr;
  if (slots && rel_sorted && (reinterpret_cast<uintptr_t>(slots) & 15) == 0) {     // {int32 spread flag, 12 bytes, slots [N, d]}
    a.spread = static_cast<const int*>(slots);
    a.slot8 = static_cast<const uint8_t*>(slots) + 16;
  }
  a.T = (int)T; a.d = (int)d; a.lpr_log2 = lanes_per_row_log2(d, 4); a.accumulate = accumulate;
  int64_t blocks = (T + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  hipLaunchKernelGGL(max_table_grad_by_type_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
