// Gene -> pathway learnable-projection pooling (reference: models/multilevel_gnn.py:212-239).
//
//   out[b, c, s, k] = sum_{g : raw_indice[b,g] = s}  x[b*NN + match[b,g], c] * [match >= 0] * W[g, k]
//
// The reference gathers [B,G,C], repeats it k times, permutes and reduces with an atomic
// scatter_reduce (>= 3 materialisations of [B,G,C,k]).  Here members are grouped by segment (forward,
// weight gradient) or by node (input gradient) on the host once per membership table, and every
// kernel is the same atomic-free row-per-wavefront gather-reduce as the CSR aggregation:
// lanes hold channels (16-byte loads), lane groups take members round-robin, shuffles merge groups.
//
// Layouts: x [R, C] (R = B*NN node rows), member tables int32 over flat members f = b*G + g,
// out_t / gout_t [B*S, K, C] (channel-contiguous; the host permutes the 30 MB result to [B,C,S,K]).
// HBM-bound: algorithmic bytes = M*C*4 (gathered rows) + M*(4+4+4K) (tables) + B*S*K*C*4.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

constexpr int kPUnroll = 4;

struct ProjArgs {                  // x / gout_t / out are T (fp32 or bf16 storage); weights and their partials fp32
  const void* x; const float* w; const void* gout_t;
  const int* ptr; const int* mem; const int* mem_row; const int* mem_seg;
  void* out; float* gw_partial;
  int rows; int C; int G; int lpr_log2;
  // row of out_t / gout_t that (segment row r = b * S + s, k) lives in.  n_groups = 0: r * K + k  ([B, S, K, C]).
  // n_groups = NG > 0 ("pooled" layout [B, NG * K, S / NG, C], s = p * NG + o): the rows of one (group o, column k) of a
  // sample are its S / NG pathways in order -- the [B', 146, C] batch the DiffPool levels consume (vae.py:238-243,
  // mlgnn/workload.py), written directly instead of through a transposing copy of the result (and of its gradient)
  int S; int n_groups;
};

template <int K>
__device__ __forceinline__ size_t proj_out_row(const ProjArgs& a, int r, int k) {
  if (a.n_groups <= 0) return (size_t)r * K + k;
  const int b = r / a.S, s = r - b * a.S;
  const int p = s / a.n_groups, o = s - p * a.n_groups;
  return ((size_t)(b * a.n_groups + o) * K + k) * (a.S / a.n_groups) + p;
}

// ---- forward: one wave per (batch, segment); out_t[seg, k, :] = sum_m x[row(m), :] * W[g(m), k]
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) void segment_project_fwd_kernel(const ProjArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << a.lpr_log2;
  const int groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2;
  const int cl = lane & (lpr - 1);
  const RowWalk walk = make_row_walk(a.rows);
  for (int cbase = 0; cbase < a.C; cbase += lpr * VEC) {
    const int c0 = cbase + cl * VEC;
    const bool cact = c0 < a.C;
    for (int r = walk.first; r < walk.r_end; r += walk.stride) {
      const int beg = a.ptr[r], end = a.ptr[r + 1];
      float acc[K][VEC];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;
      for (int base = beg; base < end; base += kWave) {
        const int cnt = min(kWave, end - base);
        int my_row = -1;
        float my_w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) my_w[k] = 0.f;
        if (lane < cnt) {
          const int f = a.mem[base + lane];
          my_row = a.mem_row[f];
          const int g = f % a.G;
#pragma unroll
          for (int k = 0; k < K; ++k) my_w[k] = a.w[(size_t)g * K + k];
        }
        for (int kk = 0; kk < cnt; kk += groups * kPUnroll) {
          float xv[kPUnroll][VEC], wk[kPUnroll][K];
#pragma unroll
          for (int u = 0; u < kPUnroll; ++u) {
            const int idx = kk + u * groups + sub;
            const int src = idx & (kWave - 1);
            const int row = __shfl(my_row, src);
#pragma unroll
            for (int k = 0; k < K; ++k) wk[u][k] = __shfl(my_w[k], src);
#pragma unroll
            for (int i = 0; i < VEC; ++i) xv[u][i] = 0.f;
            if (idx < cnt && cact && row >= 0) load_t<T, VEC>(xv[u], static_cast<const T*>(a.x) + (size_t)row * a.C + c0);
          }
#pragma unroll
          for (int u = 0; u < kPUnroll; ++u)
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc[k][i] = fmaf(xv[u][i], wk[u][k], acc[k][i]);
        }
      }
      for (int off = lpr; off < kWave; off <<= 1)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[k][i] += __shfl_xor(acc[k][i], off);
      if (sub == 0 && cact) {
#pragma unroll
        for (int k = 0; k < K; ++k) store_t<T, VEC>(static_cast<T*>(a.out) + proj_out_row<K>(a, r, k) * a.C + c0, acc[k]);
      }
    }
  }
}

// ---- input gradient, WIDE rows (>= 128 channels: one or two lane groups per wave): one wave per node row, its lane groups
// take the row's members round robin (the form of rounds 1-3; the multi-row walk below measured slower here: 326 vs 184 us
// at BASELINE configs[1], where a unit of 4 rows per group runs to the longest of its rows) ----
// one wave per node row; gx[row, :] = sum_{m -> row} sum_k gout_t[seg(m), k, :] * W[g(m), k]
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) void segment_project_bwd_x_wave_kernel(const ProjArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << a.lpr_log2;
  const int groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2;
  const int cl = lane & (lpr - 1);
  const RowWalk walk = make_row_walk(a.rows);
  for (int cbase = 0; cbase < a.C; cbase += lpr * VEC) {
    const int c0 = cbase + cl * VEC;
    const bool cact = c0 < a.C;
    for (int r = walk.first; r < walk.r_end; r += walk.stride) {
      const int beg = a.ptr[r], end = a.ptr[r + 1];
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      for (int base = beg; base < end; base += kWave) {
        const int cnt = min(kWave, end - base);
        int my_seg = 0;
        float my_w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) my_w[k] = 0.f;
        if (lane < cnt) {
          const int f = a.mem[base + lane];
          my_seg = a.mem_seg[f];
          const int g = f % a.G;
#pragma unroll
          for (int k = 0; k < K; ++k) my_w[k] = a.w[(size_t)g * K + k];
        }
        for (int kk = 0; kk < cnt; kk += groups) {
          const int idx = kk + sub;
          const int src = idx & (kWave - 1);
          const int seg = __shfl(my_seg, src);
          float wk[K];
#pragma unroll
          for (int k = 0; k < K; ++k) wk[k] = __shfl(my_w[k], src);
          if (idx < cnt && cact) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              float gv[VEC];
              load_t<T, VEC>(gv, static_cast<const T*>(a.gout_t) + proj_out_row<K>(a, seg, k) * a.C + c0);
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc[i] = fmaf(gv[i], wk[k], acc[i]);
            }
          }
        }
      }
      for (int off = lpr; off < kWave; off <<= 1)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off);
      if (sub == 0 && cact) store_t<T, VEC>(static_cast<T*>(a.out) + (size_t)r * a.C + c0, acc);
    }
  }
}

// ---- input gradient, WIDE rows, units of eight node rows (round 4) ----
// The wave-per-row form above waits out four dependent memory latencies per ROW (row pointer -> member -> segment /
// weights -> cotangent rows; 277 us at BASELINE configs[1] for a 328 MB write).  Here a wave takes a unit of eight
// consecutive rows and fetches the membership data of the whole unit in one shot -- lane = (row slot, member slot): eight
// rows x eight members per round, 2.5 members per row on average -- then walks the member slots with the gathers of all
// its rows in flight: three metadata round trips and about six gather rounds per EIGHT rows.
template <typename T, int VEC, int K, int GROUPS>
__device__ __forceinline__ void segment_project_bwd_x_unit_body(const ProjArgs& a) {
  constexpr int kU = 8;                                        // rows per unit = member slots per round
  constexpr int kMine = kU / GROUPS;                           // rows of the unit this lane group accumulates
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = kWave / GROUPS;
  const int sub = lane / lpr, cl = lane - sub * lpr;
  const int us = lane >> 3, ms = lane & 7;                     // metadata phase: row slot, member slot
  const RowWalk walk = make_row_walk((a.rows + kU - 1) / kU);
  const uint32_t kstride = a.n_groups > 0 ? (uint32_t)(a.S / a.n_groups) : 1u;      // cotangent rows of one segment: k apart
  const T* GO = static_cast<const T*>(a.gout_t);
  for (int cbase = 0; cbase < a.C; cbase += lpr * VEC) {
    const int c0 = cbase + cl * VEC;
    const bool cact = c0 < a.C;
    for (int unit = walk.first; unit < walk.r_end; unit += walk.stride) {
      const int mrow = unit * kU + us;
      int beg = 0, cnt = 0;
      if (mrow < a.rows) {
        beg = a.ptr[mrow];
        cnt = a.ptr[mrow + 1] - beg;
      }
      int longest = cnt;
#pragma unroll
      for (int off = 8; off < kWave; off <<= 1) longest = max(longest, __shfl_xor(longest, off));
      float acc[kMine][VEC];
#pragma unroll
      for (int p = 0; p < kMine; ++p)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[p][i] = 0.f;
      for (int mb = 0; mb < longest; mb += kU) {               // (one round unless a row has more than eight members)
        uint32_t obase = 0;
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = 0.f;
        if (mb + ms < cnt) {
          const int f = a.mem[beg + mb + ms];
          obase = (uint32_t)proj_out_row<K>(a, a.mem_seg[f], 0);
          const int g = f % a.G;
#pragma unroll
          for (int k = 0; k < K; ++k) w[k] = a.w[(size_t)g * K + k];
        }
        const int slots = min(kU, longest - mb);
        for (int m = 0; m < slots; ++m) {                      // (uniform over the wave)
          float gv[kMine][K][VEC], wk[kMine][K];
#pragma unroll
          for (int p = 0; p < kMine; ++p) {
            const int src = (p * GROUPS + sub) * kU + m;       // the lane that holds member m of this group's p-th row
            const uint32_t ob = (uint32_t)__shfl((int)obase, src);
            const bool on = mb + m < __shfl(cnt, src) && cact;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              wk[p][k] = __shfl(w[k], src);
#pragma unroll
              for (int i = 0; i < VEC; ++i) gv[p][k][i] = 0.f;
              if (on) load_t<T, VEC>(gv[p][k], GO + (size_t)(ob + (uint32_t)k * kstride) * a.C + c0);
            }
          }
#pragma unroll
          for (int p = 0; p < kMine; ++p)
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc[p][i] = fmaf(gv[p][k][i], wk[p][k], acc[p][i]);
        }
      }
#pragma unroll
      for (int p = 0; p < kMine; ++p) {
        const int r = unit * kU + p * GROUPS + sub;
        if (r < a.rows && cact) store_t<T, VEC>(static_cast<T*>(a.out) + (size_t)r * a.C + c0, acc[p]);
      }
    }
  }
}

// (GROUPS = lane groups per wave: 2 at 128 fp32 channels, 1 at 256)
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) void segment_project_bwd_x_unit2_kernel(const ProjArgs a) {
  segment_project_bwd_x_unit_body<T, VEC, K, 2>(a);
}
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) void segment_project_bwd_x_unit1_kernel(const ProjArgs a) {
  segment_project_bwd_x_unit_body<T, VEC, K, 1>(a);
}

// ---- input gradient, NARROW rows: gx[row, :] = sum_{m -> row} sum_k gout_t[seg(m), k, :] * W[g(m), k]
// A node has 2.5 memberships on average (G = 25 000 over 10 000 genes), so the work per row is one short chain of
// dependent loads (row pointer -> member -> segment / weights -> K cotangent rows) and one 512-byte store: a wave that
// walks ONE row at a time (rounds 1-3: 184 us for a 328 MB write) waits out four memory latencies per row.  Here every
// lane group owns a row of its own and a wave keeps kXRows of them per group in flight -- the pointer loads of all of
// them, then the j-th member of all of them, then their K * kXRows gathers -- and walks units of kXRows * groups
// consecutive rows (XCD-aware like the other row walks), so the stores of a unit are one contiguous block.
constexpr int kXRows = 4;
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) void segment_project_bwd_x_kernel(const ProjArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << a.lpr_log2;
  const int groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2;
  const int cl = lane & (lpr - 1);
  const int rows_per_unit = kXRows * groups;
  const RowWalk walk = make_row_walk((a.rows + rows_per_unit - 1) / rows_per_unit);
  for (int cbase = 0; cbase < a.C; cbase += lpr * VEC) {
    const int c0 = cbase + cl * VEC;
    const bool cact = c0 < a.C;
    for (int unit = walk.first; unit < walk.r_end; unit += walk.stride) {
      int row[kXRows], beg[kXRows], cnt[kXRows];
#pragma unroll
      for (int u = 0; u < kXRows; ++u) {
        row[u] = unit * rows_per_unit + u * groups + sub;
        const bool valid = row[u] < a.rows;
        beg[u] = valid ? a.ptr[row[u]] : 0;
        cnt[u] = valid ? a.ptr[row[u] + 1] - beg[u] : 0;
      }
      float acc[kXRows][VEC];
#pragma unroll
      for (int u = 0; u < kXRows; ++u)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[u][i] = 0.f;
      for (int j = 0;; ++j) {
        bool more = false;
#pragma unroll
        for (int u = 0; u < kXRows; ++u) more |= j < cnt[u];
        if (!__any(more)) break;                                  // (wave-uniform: every row of the unit is done)
        int f[kXRows];
#pragma unroll
        for (int u = 0; u < kXRows; ++u) f[u] = j < cnt[u] ? a.mem[beg[u] + j] : -1;
        int seg[kXRows];
        float wk[kXRows][K];
#pragma unroll
        for (int u = 0; u < kXRows; ++u) {
          seg[u] = 0;
#pragma unroll
          for (int k = 0; k < K; ++k) wk[u][k] = 0.f;
          if (f[u] >= 0) {
            seg[u] = a.mem_seg[f[u]];
            const int g = f[u] % a.G;
#pragma unroll
            for (int k = 0; k < K; ++k) wk[u][k] = a.w[(size_t)g * K + k];
          }
        }
        float gv[kXRows][K][VEC];
#pragma unroll
        for (int u = 0; u < kXRows; ++u)
#pragma unroll
          for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) gv[u][k][i] = 0.f;
            if (f[u] >= 0 && cact)
              load_t<T, VEC>(gv[u][k], static_cast<const T*>(a.gout_t) + proj_out_row<K>(a, seg[u], k) * a.C + c0);
          }
#pragma unroll
        for (int u = 0; u < kXRows; ++u)
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[u][i] = fmaf(gv[u][k][i], wk[u][k], acc[u][i]);
      }
#pragma unroll
      for (int u = 0; u < kXRows; ++u)
        if (row[u] < a.rows && cact) store_t<T, VEC>(static_cast<T*>(a.out) + (size_t)row[u] * a.C + c0, acc[u]);
    }
  }
}

// ---- weight gradient: one wave per (batch, segment); gw_partial[f, k] = <x[row(f), :], gout_t[seg, k, :]>
// (summed over the batch index by the caller).  Requires C <= 64*VEC (one channel chunk per wave).
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) void segment_project_bwd_w_kernel(const ProjArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << a.lpr_log2;
  const int groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2;
  const int cl = lane & (lpr - 1);
  const RowWalk walk = make_row_walk(a.rows);
  const int c0 = cl * VEC;
  const bool cact = c0 < a.C;
  for (int r = walk.first; r < walk.r_end; r += walk.stride) {
    const int beg = a.ptr[r], end = a.ptr[r + 1];
    float gk[K][VEC];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) gk[k][i] = 0.f;
      if (cact && end > beg) load_t<T, VEC>(gk[k], static_cast<const T*>(a.gout_t) + proj_out_row<K>(a, r, k) * a.C + c0);
    }
    for (int base = beg; base < end; base += kWave) {
      const int cnt = min(kWave, end - base);
      int my_row = -1, my_f = 0;
      if (lane < cnt) { my_f = a.mem[base + lane]; my_row = a.mem_row[my_f]; }
      // kPUnroll member rows per lane group in flight (one at a time left the gather latency-bound: 209 us for 0.65 GB)
      for (int kk = 0; kk < cnt; kk += groups * kPUnroll) {
        float xv[kPUnroll][VEC];
        int fs[kPUnroll];
        bool oks[kPUnroll];
#pragma unroll
        for (int u = 0; u < kPUnroll; ++u) {
          const int idx = kk + u * groups + sub;
          const int src = idx & (kWave - 1);
          const int row = __shfl(my_row, src);
          fs[u] = __shfl(my_f, src);
          oks[u] = idx < cnt;
#pragma unroll
          for (int i = 0; i < VEC; ++i) xv[u][i] = 0.f;
          if (oks[u] && cact && row >= 0) load_t<T, VEC>(xv[u], static_cast<const T*>(a.x) + (size_t)row * a.C + c0);
        }
#pragma unroll
        for (int u = 0; u < kPUnroll; ++u) {
          float dot[K];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) s = fmaf(xv[u][i], gk[k][i], s);
            for (int off = 1; off < lpr; off <<= 1) s += __shfl_xor(s, off);   // within the lane group
            dot[k] = s;
          }
          if (oks[u] && cl == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) a.gw_partial[(size_t)fs[u] * K + k] = dot[k];
          }
        }
      }
    }
  }
}

#define MLGNN_PROJ_LAUNCH(KERNEL, T, VEC, K, ...)                                            \
  switch (K) {                                                                               \
    case 1: hipLaunchKernelGGL((KERNEL<T, VEC, 1>), __VA_ARGS__); break;                     \
    case 2: hipLaunchKernelGGL((KERNEL<T, VEC, 2>), __VA_ARGS__); break;                     \
    case 3: hipLaunchKernelGGL((KERNEL<T, VEC, 3>), __VA_ARGS__); break;                     \
    default: hipLaunchKernelGGL((KERNEL<T, VEC, 4>), __VA_ARGS__); break;                    \
  }
// storage type x channels per lane: 16-byte accesses (4 x fp32 / 8 x bf16) when width and alignment allow, else scalar
#define MLGNN_PROJ_DISPATCH(KERNEL, bf16, wide, K, ...)                                      \
  do {                                                                                       \
    if (bf16) {                                                                              \
      if (wide) { MLGNN_PROJ_LAUNCH(KERNEL, bf16_t, 8, K, __VA_ARGS__) }                     \
      else { MLGNN_PROJ_LAUNCH(KERNEL, bf16_t, 1, K, __VA_ARGS__) }                          \
    } else {                                                                                 \
      if (wide) { MLGNN_PROJ_LAUNCH(KERNEL, float, 4, K, __VA_ARGS__) }                      \
      else { MLGNN_PROJ_LAUNCH(KERNEL, float, 1, K, __VA_ARGS__) }                           \
    }                                                                                        \
  } while (0)

static bool p16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// n_groups = 0: the plain layout (segs_per_sample unused); otherwise whole samples of S segments, S a multiple of n_groups
static bool proj_layout_ok(int64_t n_segments, int64_t S, int64_t n_groups) {
  if (n_groups == 0) return true;
  return n_groups > 0 && S > 0 && S <= INT32_MAX && S % n_groups == 0 && n_segments % S == 0;
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_segment_project_fwd(const void* x, const float* w, const int32_t* seg_ptr,
                                         const int32_t* seg_mem, const int32_t* mem_row, void* out_t,
                                         int64_t n_segments, int64_t C, int64_t G, int64_t K,
                                         int64_t segs_per_sample, int64_t n_groups, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (n_segments < 0 || C <= 0 || G <= 0 || K < 1 || K > 4 || n_segments > INT32_MAX) return MLGNN_E_SHAPE;
  if (!proj_layout_ok(n_segments, segs_per_sample, n_groups)) return MLGNN_E_SHAPE;
  if (n_segments == 0) return 0;
  if (!x || !w || !seg_ptr || !mem_row || !out_t) return MLGNN_E_NULL;
  ProjArgs a{};
  a.x = x; a.w = w; a.ptr = seg_ptr; a.mem = seg_mem; a.mem_row = mem_row;
  a.out = out_t; a.rows = (int)n_segments; a.C = (int)C; a.G = (int)G;
  a.S = (int)segs_per_sample; a.n_groups = (int)n_groups;
  const dim3 grid(grid_for_rows(n_segments)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  const bool bf16 = dtype == MLGNN_DTYPE_BF16;
  const int vec = bf16 ? 8 : 4;
  const bool wide = C % vec == 0 && p16(x) && p16(out_t);
  a.lpr_log2 = lanes_per_row_log2(C, wide ? vec : 1);
  MLGNN_PROJ_DISPATCH(segment_project_fwd_kernel, bf16, wide, (int)K, grid, block, 0, s, a);
  return (int)hipGetLastError();
}

extern "C" int mlgnn_segment_project_bwd(const void* gout_t, const void* x, const float* w,
                                         const int32_t* seg_ptr, const int32_t* seg_mem,
                                         const int32_t* mem_row, const int32_t* mem_seg,
                                         const int32_t* node_ptr, const int32_t* node_mem,
                                         void* grad_x, float* gw_partial,
                                         int64_t n_segments, int64_t n_rows, int64_t C, int64_t G, int64_t K,
                                         int64_t segs_per_sample, int64_t n_groups, int dtype, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (n_segments < 0 || n_rows < 0 || C <= 0 || G <= 0 || K < 1 || K > 4 || n_rows > INT32_MAX ||
      n_segments > INT32_MAX) return MLGNN_E_SHAPE;
  if (!proj_layout_ok(n_segments, segs_per_sample, n_groups)) return MLGNN_E_SHAPE;
  if (!gout_t || !w) return MLGNN_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 block(kBlock);
  const bool bf16 = dtype == MLGNN_DTYPE_BF16;
  const int vec = bf16 ? 8 : 4;
  const bool wide = (C % vec == 0) && p16(gout_t) && (!x || p16(x)) && (!grad_x || p16(grad_x));
  if (grad_x && n_rows > 0) {
    if (!node_ptr || !mem_seg) return MLGNN_E_NULL;
    ProjArgs a{};
    a.gout_t = gout_t; a.w = w; a.ptr = node_ptr; a.mem = node_mem; a.mem_seg = mem_seg;
    a.out = grad_x; a.rows = (int)n_rows; a.C = (int)C; a.G = (int)G;
    a.S = (int)segs_per_sample; a.n_groups = (int)n_groups;
    const dim3 grid(grid_for_rows(n_rows));
    a.lpr_log2 = lanes_per_row_log2(C, wide ? vec : 1);
    // four or more lane groups per wave (<= 64 fp32 channels): every group walks rows of its own; wider rows: a wave per row
    // (wider: units of eight rows with their membership data fetched in one shot; MLGNN_PROJ_UNIT=0: the wave-per-row form)
    static const bool unit_on = [] { const char* e = getenv("MLGNN_PROJ_UNIT"); return !(e && e[0] == '0'); }();
    const int lane_groups = kWave >> a.lpr_log2;
    const dim3 ugrid(grid_for_rows((n_rows + 7) / 8));
    if (lane_groups >= 4) MLGNN_PROJ_DISPATCH(segment_project_bwd_x_kernel, bf16, wide, (int)K, grid, block, 0, s, a);
    else if (unit_on && lane_groups == 2) MLGNN_PROJ_DISPATCH(segment_project_bwd_x_unit2_kernel, bf16, wide, (int)K, ugrid, block, 0, s, a);
    else if (unit_on && lane_groups == 1 && K <= 2) MLGNN_PROJ_DISPATCH(segment_project_bwd_x_unit1_kernel, bf16, wide, (int)K, ugrid, block, 0, s, a);
    else MLGNN_PROJ_DISPATCH(segment_project_bwd_x_wave_kernel, bf16, wide, (int)K, grid, block, 0, s, a);
    const int err = (int)hipGetLastError();
    if (err) return err;
  }
  if (gw_partial && n_segments > 0) {
    if (!x || !seg_ptr || !mem_row) return MLGNN_E_NULL;
    if (C > (wide ? 64 * vec : 64)) return MLGNN_E_SHAPE;      // one channel chunk per wave
    ProjArgs a{};
    a.gout_t = gout_t; a.x = x; a.ptr = seg_ptr; a.mem = seg_mem;
    a.mem_row = mem_row; a.gw_partial = gw_partial; a.rows = (int)n_segments; a.C = (int)C; a.G = (int)G;
    a.S = (int)segs_per_sample; a.n_groups = (int)n_groups;
    const dim3 grid(grid_for_rows(n_segments));
    a.lpr_log2 = lanes_per_row_log2(C, wide ? vec : 1);
    MLGNN_PROJ_DISPATCH(segment_project_bwd_w_kernel, bf16, wide, (int)K, grid, block, 0, s, a);
    const int err = (int)hipGetLastError();
    if (err) return err;
  }
  return 0;
}
