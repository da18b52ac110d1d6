// Shared pieces of the CSR aggregation kernels (forward: aggregate_fwd.hip, backward: aggregate_bwd.hip).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

constexpr int kUnroll = 4;                 // neighbour rows in flight per lane group and batch
// rows per wave of the forward's chunked row walk (common.h).  Measured at config 2 (64 x 10k-node graphs):
// 4 rows per wave is best (2 / 4 / 8: softmax 0.65 / 0.62 / 0.64 ms, mean 0.41 / 0.41 / 0.42 ms; the strided
// persistent walk: 0.73 / 0.51 ms).  The backward keeps the strided walk: it is bound
// by per-edge instruction issue, and one edge-term partial per workgroup favours few, long-lived workgroups.
constexpr int kFwdRowsPerWave = 4;
constexpr int kNoCap = 0x3fffffff;
constexpr int kHubBlocks = 256;             // workgroups of the backward launch over the extra chunks of long rows
constexpr float kPowLo = 1e-7f, kPowHi = 1e1f;   // torch_message.py:69
constexpr float kNegBig = -3.0e38f;

// M_GEN_RANK{1,2,4,8}: edge term  e_e[c] = sum_k a_e[k] * U[k][c] + v[c]  with r (padded to 1/2/4/8) raw edge
// attributes per edge -- the Linear(r -> H) / Linear(H -> d) encoder stack kept factored
enum Mode { M_IDENTITY = 0, M_WEIGHTED = 1, M_GEN_NONE = 2, M_GEN_RANK1 = 3, M_GEN_FULL = 4,
            M_GEN_RANK2 = 5, M_GEN_RANK4 = 6, M_GEN_RANK8 = 7 };

template <int MODE>
__host__ __device__ constexpr int rank_of() {
  return MODE == M_GEN_RANK1 ? 1 : MODE == M_GEN_RANK2 ? 2 : MODE == M_GEN_RANK4 ? 4 : MODE == M_GEN_RANK8 ? 8 : 0;
}
inline int rank_of_mode(int mode) {
  return mode == M_GEN_RANK1 ? 1 : mode == M_GEN_RANK2 ? 2 : mode == M_GEN_RANK4 ? 4 : mode == M_GEN_RANK8 ? 8 : 0;
}
// scalars per edge the kernels read from `ew`: the weight (M_WEIGHTED) or the padded attribute row
template <int MODE>
__host__ __device__ constexpr int edge_scalars() { return MODE == M_WEIGHTED ? 1 : rank_of<MODE>(); }
enum Aggr { A_SUM = 0, A_MAX = 2, A_SOFTMAX = 3, A_POWER = 4 };   // MEAN = SUM + epilogue flag

template <int V> using IC = std::integral_constant<int, V>;
template <typename T> struct TypeTag { using type = T; };
template <bool V> using BC = std::integral_constant<bool, V>;

struct FwdArgs {                      // x / efull / out are T (fp32 or bf16); everything else fp32 / int32
  const void* x; const int* rowptr; const int* col;
  const float* ew; const float* eu; const float* ev; const void* efull; const int* eid;
  void* out; float* aux; float* aux2; int* argmax; float* rowmax;
  const float* t_dev; const float* p_dev;
  int N; int d; int lpr_log2; int mean; int add_root;
  float t; float p; float eps;
  // long rows (csrc/hub.hip): every row is clamped to its first `cap` edges (kNoCap: off; a clamped row's result is
  // a partial: no root term).  The VIRT instantiation of the kernel walks the extra chunks instead:
  // vrows[r] = {real row, first edge, one past the last edge}, *vcount of them, results to scratch row r (fp32)
  int cap; const int* vrows; const int* vcount;
};

// t / p either immediate or read from device memory (learnable parameters: no host sync)
struct Scalars { float t, t_log2e, p; };
__device__ __forceinline__ Scalars read_scalars(const float* t_dev, const float* p_dev, float t, float p) {
  Scalars s;
  s.t = t_dev ? t_dev[0] : t;
  s.p = p_dev ? p_dev[0] : p;
  s.t_log2e = s.t * kLog2e;
  return s;
}

template <int MODE>
__device__ __forceinline__ constexpr bool is_gen() { return MODE >= M_GEN_NONE; }

// rows of x / out / aux are addressed with 32-bit byte offsets from a uniform base (tensors < 4 GiB,
// checked on the host): one v_mul + v_add per gathered row instead of 64-bit multiply-adds
template <typename T, int VEC>
__device__ __forceinline__ void load_row(float (&r)[VEC], const T* base, uint32_t byte_off) {
  load_t<T, VEC>(r, reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off));
}
template <int VEC>
__device__ __forceinline__ void load_row(int (&r)[VEC], const int* base, uint32_t byte_off) {
  load_vec<VEC>(r, reinterpret_cast<const int*>(reinterpret_cast<const char*>(base) + byte_off));
}
// WIDE instantiations ([N,d] tensors of 4 GiB and more -- 288 GB of HBM hold graphs of tens of millions of nodes): the
// lanes pass the ROW INDEX around and every gathered row pays a 64-bit multiply-add for its address; chosen on the host
// only when N * d * 4 does not fit 32 bits.
template <typename T, int VEC>
__device__ __forceinline__ void load_row(float (&r)[VEC], const T* base, uint64_t byte_off) {
  load_t<T, VEC>(r, reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off));
}
template <int VEC>
__device__ __forceinline__ void load_row(int (&r)[VEC], const int* base, uint64_t byte_off) {
  load_vec<VEC>(r, reinterpret_cast<const int*>(reinterpret_cast<const char*>(base) + byte_off));
}
// row key a lane keeps for its edge (what is shuffled to the lane groups) and the byte offset of the row chunk from it
template <bool WIDE>
__device__ __forceinline__ uint32_t row_key(int row, uint32_t row_bytes) { return WIDE ? (uint32_t)row : (uint32_t)row * row_bytes; }
template <bool WIDE>
__device__ __forceinline__ auto row_offset(uint32_t key, uint32_t row_bytes, uint32_t c_bytes) {
  if constexpr (WIDE) return (uint64_t)key * row_bytes + c_bytes;
  else return key + c_bytes;
}
// does an [N, d] tensor need the WIDE kernels?  (fp32 side arrays -- lse, argmax -- share the row index: 4 bytes / element)
inline bool needs_wide_rows(int64_t N, int64_t d) { return N * d * 4 >= ((int64_t)1 << 32); }
inline bool force_wide_rows() {                      // tests: the WIDE kernels on small inputs
  static const int v = [] { const char* e = getenv("MLGNN_FORCE_WIDE"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}

// the ES scalars of edge slot e (row e of the [E, ES] table; rows are ES*4-byte aligned)
template <int ES>
__device__ __forceinline__ void load_edge_scalars(float* dst, const float* __restrict__ table, size_t e) {
  if constexpr (ES == 1) dst[0] = table[e];
  else if constexpr (ES == 2) {
    const float2 v = *reinterpret_cast<const float2*>(table + e * 2);
    dst[0] = v.x; dst[1] = v.y;
  } else {
#pragma unroll
    for (int q = 0; q < ES; q += 4) {
      const float4 v = *reinterpret_cast<const float4*>(table + e * ES + q);
      dst[q] = v.x; dst[q + 1] = v.y; dst[q + 2] = v.z; dst[q + 3] = v.w;
    }
  }
}

// pre-activation z of the GEN message for one channel; a[] = the edge's scalars, u[] = column c of U
template <int MODE>
__device__ __forceinline__ float pre_act(float xj, const float* a, const float* u, float v, float ef) {
  if constexpr (rank_of<MODE>() > 0) {
    float e = v;
#pragma unroll
    for (int k = 0; k < rank_of<MODE>(); ++k) e = fmaf(a[k], u[k], e);
    return xj + e;
  } else if constexpr (MODE == M_GEN_FULL) return xj + ef;
  else return xj;
}

// Non-finite inputs.  relu / max / the power clamp as v_max / v_min return the OTHER operand when one is NaN: a NaN in
// x_j or in the edge term would silently vanish from the aggregate, where the reference carries it to the loss
// (relu(NaN) = NaN, torch_vertex.py:94-101).  gfx950 has the IEEE-754-2019 forms (v_maximum3_f32 / v_minimum3_f32:
// NaN if any operand is NaN), so the GEN relu and the power clamp keep a NaN at no extra cost; +-Inf follow torch too
// (relu(+Inf) = +Inf, relu(-Inf) = 0) and the sums / exponentials carry them to the row's result by themselves.
// Inline asm: the aggregation translation units are built with -fno-honor-nans.
__device__ __forceinline__ float relu_nan(float z) {
  float m;
  asm("v_maximum3_f32 %0, %1, 0, 0" : "=v"(m) : "v"(z));
  return m;
}
__device__ __forceinline__ float clamp_nan(float x, float lo, float hi) {
  float t, r;
  asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(t) : "v"(x), "v"(lo));
  asm("v_minimum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(t), "v"(hi));
  return r;
}
template <int MODE, bool ADD_EPS>
__device__ __forceinline__ float message(float xj, const float* a, const float* u, float v, float ef, float eps) {
  if constexpr (MODE == M_IDENTITY) return xj;
  else if constexpr (MODE == M_WEIGHTED) return xj * a[0];
  else if constexpr (ADD_EPS) return fmaxf(pre_act<MODE>(xj, a, u, v, ef), 0.0f) + eps;
  else return fmaxf(pre_act<MODE>(xj, a, u, v, ef), 0.0f);
}

// The messages of one gathered row chunk (VEC channels of one neighbour), two channels per instruction where the width
// allows: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 run two fp32 lanes per VALU slot and these kernels are bound by
// VALU issue (SQ_ACTIVE_INST_VALU 97 % of the cycles at config 1).  z (GEN modes): the pre-activation, for the backward.
using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int MODE, int VEC, int ESA, bool KEEP_NAN, bool ADD_EPS>
__device__ __forceinline__ void row_messages(float (&m)[VEC], float (&z)[VEC], const float (&xv)[VEC], const float (&wa)[ESA],
                                             const float (&eu)[VEC][ESA], const float (&ev)[VEC], const float (&ef)[VEC],
                                             float eps) {
  if constexpr (VEC % 2 == 0) {
#pragma unroll
    for (int i = 0; i < VEC; i += 2) {
      f32x2 x2 = {xv[i], xv[i + 1]};
      if constexpr (MODE == M_IDENTITY) {
        m[i] = x2.x; m[i + 1] = x2.y;
        z[i] = x2.x; z[i + 1] = x2.y;
      } else if constexpr (MODE == M_WEIGHTED) {
        const f32x2 w2 = {wa[0], wa[0]};
        x2 = x2 * w2;
        m[i] = x2.x; m[i + 1] = x2.y;
        z[i] = x2.x; z[i + 1] = x2.y;
      } else {
        if constexpr (rank_of<MODE>() > 0) {
          f32x2 e2 = {ev[i], ev[i + 1]};
#pragma unroll
          for (int k = 0; k < rank_of<MODE>(); ++k) {
            const f32x2 a2 = {wa[k], wa[k]}, u2 = {eu[i][k], eu[i + 1][k]};
            e2 = __builtin_elementwise_fma(a2, u2, e2);
          }
          x2 = x2 + e2;
        } else if constexpr (MODE == M_GEN_FULL) {
          const f32x2 e2 = {ef[i], ef[i + 1]};
          x2 = x2 + e2;
        }
        z[i] = x2.x; z[i + 1] = x2.y;
        f32x2 r2;
        r2.x = KEEP_NAN ? relu_nan(x2.x) : fmaxf(x2.x, 0.0f);
        r2.y = KEEP_NAN ? relu_nan(x2.y) : fmaxf(x2.y, 0.0f);
        if constexpr (ADD_EPS) { const f32x2 eps2 = {eps, eps}; r2 = r2 + eps2; }
        m[i] = r2.x; m[i + 1] = r2.y;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if constexpr (!is_gen<MODE>()) {
        m[i] = message<MODE, false>(xv[i], wa, eu[i], ev[i], ef[i], eps);
        z[i] = m[i];
      } else {
        z[i] = pre_act<MODE>(xv[i], wa, eu[i], ev[i], ef[i]);
        const float r = KEEP_NAN ? relu_nan(z[i]) : fmaxf(z[i], 0.0f);
        m[i] = ADD_EPS ? r + eps : r;
      }
    }
  }
}

// csrc/hub.hip
struct HubFwdArgs {
  const int* hubs; const int* vrows; const int* counts; const int* rowptr;
  void* out; float* aux; float* aux2; int* argmax; float* rowmax; const void* x;      // real rows
  const void* out_v; const float* aux_v; const float* aux2_v; const int* argmax_v;    // extra chunks
  const float* p_dev; float p;
  int d; int cap; int aggr; int mean; int add_root; int second;
};
int hub_combine_fwd(const HubFwdArgs& a, bool bf16, hipStream_t s);
int hub_combine_bwd(const int* hubs, const int* counts, void* gx, const void* gx_v, int d, bool bf16, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
inline int pick_mode(int msg, int edge_mode, int edge_rank) {
  if (msg == MLGNN_MSG_IDENTITY) return M_IDENTITY;
  if (msg == MLGNN_MSG_WEIGHTED) return M_WEIGHTED;
  if (msg == MLGNN_MSG_GEN) {
    if (edge_mode == MLGNN_EDGE_NONE) return M_GEN_NONE;
    if (edge_mode == MLGNN_EDGE_RANK1)
      return edge_rank == 1 ? M_GEN_RANK1 : edge_rank == 2 ? M_GEN_RANK2 : edge_rank == 4 ? M_GEN_RANK4
             : edge_rank == 8 ? M_GEN_RANK8 : -1;
    if (edge_mode == MLGNN_EDGE_FULL) return M_GEN_FULL;
  }
  return -1;
}

inline int pick_aggr(int aggr) {
  switch (aggr) {
    case MLGNN_AGGR_SUM: case MLGNN_AGGR_MEAN: return A_SUM;
    case MLGNN_AGGR_MAX: return A_MAX;
    case MLGNN_AGGR_SOFTMAX: return A_SOFTMAX;
    case MLGNN_AGGR_POWER: return A_POWER;
  }
  return -1;
}

inline bool is_gen_mode(int mode) { return mode >= M_GEN_NONE; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// f(IC<MODE>, IC<AGGR>) for the valid (mode, aggregator) pairs
template <int MODE, typename F>
inline void for_aggr(int ag, F&& f) {
  switch (ag) {
    case A_SUM: f(IC<MODE>{}, IC<A_SUM>{}); break;
    case A_MAX: f(IC<MODE>{}, IC<A_MAX>{}); break;
    case A_SOFTMAX: f(IC<MODE>{}, IC<A_SOFTMAX>{}); break;
    default: f(IC<MODE>{}, IC<A_POWER>{}); break;
  }
}
template <typename F>
inline void for_mode_aggr(int mode, int ag, F&& f) {
  switch (mode) {
    case M_IDENTITY: f(IC<M_IDENTITY>{}, IC<A_SUM>{}); break;
    case M_WEIGHTED: f(IC<M_WEIGHTED>{}, IC<A_SUM>{}); break;
    case M_GEN_NONE: for_aggr<M_GEN_NONE>(ag, f); break;
    case M_GEN_RANK1: for_aggr<M_GEN_RANK1>(ag, f); break;
    case M_GEN_RANK2: for_aggr<M_GEN_RANK2>(ag, f); break;
    case M_GEN_RANK4: for_aggr<M_GEN_RANK4>(ag, f); break;
    case M_GEN_RANK8: for_aggr<M_GEN_RANK8>(ag, f); break;
    case M_GEN_FULL: for_aggr<M_GEN_FULL>(ag, f); break;
    default: break;
  }
}

}  // namespace mlgnn
