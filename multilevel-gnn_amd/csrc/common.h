// Shared device helpers for the gfx950 kernels of libmlgnn.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlgnn {

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kBlock = 256;        // 4 waves per workgroup, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kXcds = 8;           // MI355X: 8 XCDs, each with its own L2
constexpr int kMaxBlocks = 2048;   // 256 CUs x 8 resident workgroups

template <int VEC>
__device__ __forceinline__ void load_vec(float (&r)[VEC], const float* __restrict__ p) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else if constexpr (VEC == 8) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    const float4 q = *reinterpret_cast<const float4*>(p + 4);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; r[4] = q.x; r[5] = q.y; r[6] = q.z; r[7] = q.w;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = p[i];
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec(int (&r)[VEC], const int* __restrict__ p) {
  if constexpr (VEC == 4) {
    const int4 t = *reinterpret_cast<const int4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else if constexpr (VEC == 8) {
    const int4 t = *reinterpret_cast<const int4*>(p);
    const int4 q = *reinterpret_cast<const int4*>(p + 4);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; r[4] = q.x; r[5] = q.y; r[6] = q.z; r[7] = q.w;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = p[i];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else if constexpr (VEC == 8) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(r[4], r[5], r[6], r[7]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = r[i];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(int* __restrict__ p, const int (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<int4*>(p) = make_int4(r[0], r[1], r[2], r[3]);
  } else if constexpr (VEC == 8) {
    *reinterpret_cast<int4*>(p) = make_int4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<int4*>(p + 4) = make_int4(r[4], r[5], r[6], r[7]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = r[i];
  }
}

// ---- storage types: activations are kept as fp32 or bf16 in HBM, arithmetic is always fp32 --------
struct bf16_t { uint16_t bits; };

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {         // round to nearest even (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}

template <typename T, int VEC>
__device__ __forceinline__ void load_t(float (&r)[VEC], const T* __restrict__ p) {
  if constexpr (sizeof(T) == 4) {
    load_vec<VEC>(r, reinterpret_cast<const float*>(p));
  } else if constexpr (VEC == 8) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[2 * i] = __builtin_bit_cast(float, w[i] << 16);
      r[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  }
}

template <typename T, int VEC>
__device__ __forceinline__ void store_t(T* __restrict__ p, const float (&r)[VEC]) {
  if constexpr (sizeof(T) == 4) {
    store_vec<VEC>(reinterpret_cast<float*>(p), r);
  } else if constexpr (VEC == 8) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(r[2 * i]) | ((uint32_t)f32_to_bf16(r[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) reinterpret_cast<uint16_t*>(p)[i] = f32_to_bf16(r[i]);
  }
}

// Scalar access to fp32-or-bf16 storage for the small pooled-graph kernels (densesage.hip, diffpool.hip):
// read-only view of a T array that indexes like a float array (bf16 storage is widened on the load; every product and
// sum below is fp32 either way, results are rounded once at the store)
template <typename T>
struct StoredIn {
  const T* p;
  __device__ __forceinline__ float operator[](size_t i) const {
    if constexpr (sizeof(T) == 4) return reinterpret_cast<const float*>(p)[i];
    else return bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  }
};
template <typename T>
__device__ __forceinline__ void stored_write(void* base, size_t i, float v) {
  if constexpr (sizeof(T) == 4) static_cast<float*>(base)[i] = v;
  else static_cast<uint16_t*>(base)[i] = f32_to_bf16(v);
}

// Streaming (written once, not read again by this kernel) 16-byte stores with the non-temporal hint, for kernels whose
// speed hangs on the L2 hit rate of a gather running next to the stream (the CSR aggregations).  MLGNN_NT_STORES=0
// compiles them as plain stores (same-box A/B through tools/build_variant.py).
#ifndef MLGNN_NT_STORES
#define MLGNN_NT_STORES 1
#endif
template <typename T, int VEC>
__device__ __forceinline__ void store_t_stream(T* __restrict__ p, const float (&r)[VEC]) {
#if MLGNN_NT_STORES
  using u4 = __attribute__((ext_vector_type(4))) uint32_t;
  if constexpr (sizeof(T) == 4 && VEC == 4) {
    const u4 v = {__builtin_bit_cast(uint32_t, r[0]), __builtin_bit_cast(uint32_t, r[1]),
                  __builtin_bit_cast(uint32_t, r[2]), __builtin_bit_cast(uint32_t, r[3])};
    __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p));
  } else if constexpr (sizeof(T) == 2 && VEC == 8) {
    u4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (uint32_t)f32_to_bf16(r[2 * i]) | ((uint32_t)f32_to_bf16(r[2 * i + 1]) << 16);
    __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p));
  } else {
    store_t<T, VEC>(p, r);
  }
#else
  store_t<T, VEC>(p, r);
#endif
}

// All-reduce over the 32-lane half of a wavefront without the LDS crossbar: rotations inside each row of 16 lanes
// are DPP modifiers of the VALU op (row_ror:8/4/2/1), only the last step (row <-> row) is a ds_bpermute.  For
// epilogues that reduce many values per lane (a `__shfl_xor` ladder is five crossbar trips per value).
template <int ROR>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROR, 0xf, 0xf, false));
}
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_row_ror<8>(v); v += dpp_row_ror<4>(v); v += dpp_row_ror<2>(v); v += dpp_row_ror<1>(v);
  return v + __shfl_xor(v, 16);
}
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, dpp_row_ror<8>(v)); v = fmaxf(v, dpp_row_ror<4>(v));
  v = fmaxf(v, dpp_row_ror<2>(v)); v = fmaxf(v, dpp_row_ror<1>(v));
  return fmaxf(v, __shfl_xor(v, 16));
}

// relu that carries a NaN through like torch's (fmaxf / v_max_f32 return the non-NaN operand)
__device__ __forceinline__ float relu_keep_nan(float y) { return (y < 0.f) ? 0.f : y; }

// exp2 / log2 on the transcendental unit (v_exp_f32 / v_log_f32, ~1 ulp)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// |lse| (log2 units) up to which the softmax backward takes the two factors 2^(t m) <= 2^lse and 2^(-lse) apart (the
// shifted cotangent gt = go * 2^(-lse): csrc/aggregate_bwd.hip, and its producer epilogue in csrc/tallgemm.hip):
// 2^(+-60) leaves fp32 more than 60 binades on either side for the cotangent itself
constexpr float kMaxLse = 60.0f;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Rows [r0, r1) one XCD walks, and this wave's position in that walk.  Workgroups are dealt
// round-robin over the 8 XCDs (observed, not guaranteed: it only affects speed), so block b
// belongs to group b % 8; each group sweeps one contiguous eighth of the rows with all of its
// waves side by side, which keeps the gathered neighbour rows of a graph in that XCD's L2.
struct RowWalk {
  int r_begin, r_end, first, stride;
};

__device__ __forceinline__ RowWalk make_row_walk(int n_rows) {
  const int xcd = blockIdx.x % kXcds;
  const int slot = blockIdx.x / kXcds;
  const int blocks_per_xcd = gridDim.x / kXcds;          // grid is a multiple of 8
  const int rows_per_xcd = (n_rows + kXcds - 1) / kXcds;
  RowWalk w;
  w.r_begin = xcd * rows_per_xcd;
  w.r_end = min(n_rows, w.r_begin + rows_per_xcd);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  w.first = w.r_begin + slot * kWavesPerBlock + wave;
  w.stride = blocks_per_xcd * kWavesPerBlock;
  return w;
}

// Chunked variant: workgroup `slot` of an XCD owns the contiguous chunk [slot*C, (slot+1)*C) of that XCD's
// eighth (C = rows per workgroup, a multiple of 4; wave w takes rows w, w+4, ...).  Workgroups are dispatched
// in index order, so at any time the resident workgroups of an XCD cover one window of a few thousand
// consecutive rows that advances ONCE through the eighth: the rows gathered by a graph's nodes are touched
// in one pass while that graph is the L2's working set, instead of once per generation of resident
// workgroups as with the strided persistent walk above.
__device__ __forceinline__ RowWalk make_chunk_walk(int n_rows) {
  const int xcd = blockIdx.x % kXcds;
  const int slot = blockIdx.x / kXcds;
  const int blocks_per_xcd = gridDim.x / kXcds;          // grid is a multiple of 8
  const int rows_per_xcd = (n_rows + kXcds - 1) / kXcds;
  int chunk = (rows_per_xcd + blocks_per_xcd - 1) / blocks_per_xcd;
  chunk = (chunk + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;
  RowWalk w;
  w.r_begin = xcd * rows_per_xcd;
  const int x_end = min(n_rows, w.r_begin + rows_per_xcd);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  w.first = w.r_begin + slot * chunk + wave;
  w.r_end = min(x_end, w.r_begin + (slot + 1) * chunk);
  w.stride = kWavesPerBlock;
  return w;
}

inline int grid_for_chunks(int64_t n_rows, int rows_per_wave) {
  const int64_t rows_per_xcd = (n_rows + kXcds - 1) / kXcds;
  int64_t per_xcd = (rows_per_xcd + (int64_t)kWavesPerBlock * rows_per_wave - 1) / ((int64_t)kWavesPerBlock * rows_per_wave);
  if (per_xcd < 1) per_xcd = 1;
  return (int)(per_xcd * kXcds);
}

inline int grid_for_rows(int64_t n_rows) {
  int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  blocks = (blocks + kXcds - 1) / kXcds * kXcds;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  if (blocks < kXcds) blocks = kXcds;
  return (int)blocks;
}

// lanes-per-row (power of two) for d channels at VEC floats per lane, capped at one wave
inline int lanes_per_row_log2(int64_t d, int vec) {
  int64_t need = (d + vec - 1) / vec;
  int l = 0;
  while ((1 << l) < need && l < 6) ++l;
  return l;
}

// ws[nblk][cols] -> out[cols] in a fixed summation order (defined in aggregate_bwd.hip)
// Fixed-point accumulator of a table gradient (csrc/embedding.hip, the max-aggregation backward with a table edge
// term): a 256-byte header {bits of max |cotangent|, non-finite flag, headroom bits} followed by int64 [T, d].
// Integer atomic adds commute, so the sums are independent of the order the edges arrive in (bitwise reproducible).
// The scale is the power of two that maps |v| <= max onto |v * scale| < 2^bits.
constexpr int kFixHeaderBytes = 256;
__device__ __forceinline__ int fix_scale_exponent(const uint32_t* hdr) {
  int e = (int)((hdr[0] >> 23) & 0xffu);                     // biased exponent of the maximum: |v| < 2^(e - 126)
  e = e < 1 ? 1 : e;
  const int se = (int)hdr[2] - e + 253;                     // biased exponent of 2^(bits - (e - 126))
  return se < 1 ? 1 : (se > 253 ? 253 : se);
}
__device__ __forceinline__ float fix_scale_of(const uint32_t* hdr) {
  return __builtin_bit_cast(float, (uint32_t)fix_scale_exponent(hdr) << 23);
}

// accumulate: out += the sum (a later row slab of one reduction: slab sums are added in slab order, still bitwise reproducible)
void launch_reduce_partials(const float* ws, float* out, int nblk, int cols, hipStream_t stream, bool accumulate = false);

// Row slabs of the dense kernels that address their operands with 32-bit byte offsets from a uniform base: the most rows
// (a multiple of 4096) whose widest operand row block stays below 4 GiB.  5.12 M rows x 256 fp32 columns (512 graphs of
// BASELINE configs[3] on one GPU) = 2 slabs.
inline int64_t dense_slab_rows(int64_t widest_cols) {
  const int64_t rows = (((int64_t)1 << 32) - 1) / (widest_cols * 4);
  return rows / 4096 * 4096;
}

}  // namespace mlgnn
