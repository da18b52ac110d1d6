// Long rows ("hub" nodes) of the CSR aggregation kernels.
//
// One wavefront owns one row in csr_aggregate_fwd / _bwd: a transcription factor with 15 000 targets (the cross-omics
// edges of dataloader/multiloader.py:664-671 are built per TF) keeps a single wave busy for ~0.7 ms while the rest of
// the chip has long finished.  Rows longer than `cap` edges are therefore cut into chunks of `cap`:
//   * the main launch handles the first chunk of every row (a plain clamp of the row end),
//   * a second launch of the SAME kernel walks the table of extra chunks (real row, first edge, last edge) and
//     writes per-chunk partial results to scratch rows,
//   * a combine kernel folds a row's partials in chunk order -- fixed order, no atomics, bitwise reproducible.
//     Chunk partials are kept in fp32 whatever the storage type of the activations (a bf16 partial per chunk would
//     add a rounding per chunk to sums that may cancel).
// Partial results of every aggregator combine exactly:  sum / mean by (degree-weighted) addition; max by value with
// the earlier edge winning ties; softmax through the chunks' log-sum-exp:  out = sum_v 2^(lse_v - lse) out_v;
// power through the chunk means of m^p.  The backward (one wave per SOURCE row) is a plain sum.
//
// The chunk table is built on the device (hub_rows_kernel) right after the CSR: no host round trip, and a graph
// without long rows costs two empty launches per aggregation.
#include "aggregate_common.h"

namespace mlgnn {

// one thread per row: rows longer than cap reserve their extra chunks with ONE atomicAdd (so a row's chunks are
// consecutive and the combine order is fixed, whatever order the rows arrive in)
__global__ __launch_bounds__(256) void hub_rows_kernel(const int* __restrict__ rowptr, int N, int cap, int capacity,
                                                       int* __restrict__ vrows, int* __restrict__ hubs,
                                                       int* __restrict__ counts) {
  for (int r = blockIdx.x * 256 + threadIdx.x; r < N; r += gridDim.x * 256) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    if (end - beg <= cap) continue;
    const int k = (end - beg + cap - 1) / cap - 1;
    const int v0 = atomicAdd(&counts[0], k);
    const int h = atomicAdd(&counts[1], 1);
    if (v0 + k > capacity || h >= capacity) continue;          // cannot happen for capacity >= E / cap + 1
    hubs[3 * h] = r; hubs[3 * h + 1] = v0; hubs[3 * h + 2] = k;
    for (int c = 0; c < k; ++c) {
      vrows[3 * (v0 + c)] = r;
      vrows[3 * (v0 + c) + 1] = beg + (c + 1) * cap;
      vrows[3 * (v0 + c) + 2] = min(end, beg + (c + 2) * cap);
    }
  }
}

// one wave per split row; lanes over channels
template <typename T>
__global__ __launch_bounds__(kBlock) void hub_combine_fwd_kernel(const HubFwdArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave_global = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  const int n_hubs = a.counts[1];
  T* OUT = static_cast<T*>(a.out);
  const float* OUTV = static_cast<const float*>(a.out_v);       // chunk partials: fp32 whatever T is
  const T* X = static_cast<const T*>(a.x);
  for (int h = wave_global; h < n_hubs; h += n_waves) {
    const int r = a.hubs[3 * h], v0 = a.hubs[3 * h + 1], k = a.hubs[3 * h + 2];
    const float deg = (float)(a.rowptr[r + 1] - a.rowptr[r]);
    float row_abs = 0.f;
    for (int c = lane; c < a.d; c += kWave) {
      const size_t at = (size_t)r * a.d + c;
      float o[1], t[1];
      load_t<T, 1>(o, OUT + at);
      float res = o[0], lse = 0.f, a2 = 0.f;
      int am = -1;
      if (a.aggr == A_SUM) {
        float acc = a.mean ? o[0] * (float)a.cap : o[0];
        for (int v = 0; v < k; ++v) {
          t[0] = OUTV[(size_t)(v0 + v) * a.d + c];
          acc += a.mean ? t[0] * (float)(a.vrows[3 * (v0 + v) + 2] - a.vrows[3 * (v0 + v) + 1]) : t[0];
        }
        res = a.mean ? acc / deg : acc;
      } else if (a.aggr == A_MAX) {
        am = a.argmax ? a.argmax[at] : -1;
        for (int v = 0; v < k; ++v) {
          t[0] = OUTV[(size_t)(v0 + v) * a.d + c];
          if (t[0] > res) {                                      // a later chunk only wins with a strictly larger value
            res = t[0];
            if (a.argmax) am = a.argmax_v[(size_t)(v0 + v) * a.d + c];
          }
        }
      } else if (a.aggr == A_SOFTMAX) {
        // chunk results are normalised inside their chunk; weights between chunks come from the chunks' lse (log2)
        float big = a.aux[at];
        for (int v = 0; v < k; ++v) big = fmaxf(big, a.aux_v[(size_t)(v0 + v) * a.d + c]);
        float w = fast_exp2(a.aux[at] - big);
        float s = w, acc = w * o[0];
        a2 = a.second ? w * a.aux2[at] : 0.f;
        for (int v = 0; v < k; ++v) {
          const size_t av = (size_t)(v0 + v) * a.d + c;
          t[0] = OUTV[av];
          w = fast_exp2(a.aux_v[av] - big);
          s += w;
          acc = fmaf(w, t[0], acc);
          if (a.second) a2 = fmaf(w, a.aux2_v[av], a2);
        }
        const float inv = 1.0f / s;
        res = acc * inv;
        a2 *= inv;
        lse = big + fast_log2(s);
      } else {  // A_POWER: aux = mean over the chunk of m^p (before the outer clamp), aux2 = mean of m^p ln m
        const float p = a.p_dev ? a.p_dev[0] : a.p;
        float mu = a.aux[at] * (float)a.cap;
        a2 = a.second ? a.aux2[at] * (float)a.cap : 0.f;
        for (int v = 0; v < k; ++v) {
          const size_t av = (size_t)(v0 + v) * a.d + c;
          const float n = (float)(a.vrows[3 * (v0 + v) + 2] - a.vrows[3 * (v0 + v) + 1]);
          mu = fmaf(a.aux_v[av], n, mu);
          if (a.second) a2 = fmaf(a.aux2_v[av], n, a2);
        }
        mu /= deg;
        a2 /= deg;
        lse = mu;
        const float muc = clamp_nan(mu, kPowLo, kPowHi);
        res = fast_exp2(fast_log2(muc) / p);
      }
      if (a.add_root) {
        load_t<T, 1>(t, X + at);
        res += t[0];
      }
      o[0] = res;
      store_t<T, 1>(OUT + at, o);
      if ((a.aggr == A_SOFTMAX || a.aggr == A_POWER) && a.aux) a.aux[at] = lse;
      if (a.second && a.aux2) a.aux2[at] = a2;
      if (a.aggr == A_MAX && a.argmax) a.argmax[at] = am;
      row_abs = fmaxf(row_abs, fabsf(res));
    }
    if (a.rowmax) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) row_abs = fmaxf(row_abs, __shfl_xor(row_abs, o));
      if (lane == 0) a.rowmax[r] = row_abs;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void hub_combine_bwd_kernel(const int* __restrict__ hubs, const int* __restrict__ counts,
                                                                 T* __restrict__ gx, const float* __restrict__ gx_v, int d) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave_global = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  const int n_hubs = counts[1];
  for (int h = wave_global; h < n_hubs; h += n_waves) {
    const int r = hubs[3 * h], v0 = hubs[3 * h + 1], k = hubs[3 * h + 2];
    for (int c = lane; c < d; c += kWave) {
      float acc[1], t[1];
      load_t<T, 1>(acc, gx + (size_t)r * d + c);
      for (int v = 0; v < k; ++v) {
        t[0] = gx_v[(size_t)(v0 + v) * d + c];
        acc[0] += t[0];
      }
      store_t<T, 1>(gx + (size_t)r * d + c, acc);
    }
  }
}

int hub_combine_fwd(const HubFwdArgs& a, bool bf16, hipStream_t s) {
  if (bf16) hipLaunchKernelGGL(hub_combine_fwd_kernel<bf16_t>, dim3(64), dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL(hub_combine_fwd_kernel<float>, dim3(64), dim3(kBlock), 0, s, a);
  return (int)hipGetLastError();
}

int hub_combine_bwd(const int* hubs, const int* counts, void* gx, const void* gx_v, int d, bool bf16, hipStream_t s) {
  if (bf16) hipLaunchKernelGGL(hub_combine_bwd_kernel<bf16_t>, dim3(64), dim3(kBlock), 0, s, hubs, counts, (bf16_t*)gx, (const float*)gx_v, d);
  else hipLaunchKernelGGL(hub_combine_bwd_kernel<float>, dim3(64), dim3(kBlock), 0, s, hubs, counts, (float*)gx, (const float*)gx_v, d);
  return (int)hipGetLastError();
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int64_t mlgnn_hub_capacity(int64_t E, int cap) {
  if (E < 0 || cap < 1) return MLGNN_E_SHAPE;
  return E / cap + 1;
}

extern "C" int64_t mlgnn_hub_scratch_bytes(int64_t capacity, int64_t d) {
  if (capacity < 0 || d < 0) return MLGNN_E_SHAPE;
  return capacity * d * 16 + 256;      // out (<= 4 B) + lse + second moment + argmax per channel of every extra chunk
}

extern "C" int mlgnn_hub_rows(const int32_t* rowptr, int64_t N, int cap, int64_t capacity, int32_t* vrows,
                              int32_t* hubs, int32_t* counts, void* stream) {
  if (N < 0 || N > INT32_MAX || cap < 1 || capacity < 1 || capacity > INT32_MAX / 3) return MLGNN_E_SHAPE;
  if (!rowptr || !vrows || !hubs || !counts) return MLGNN_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  int err = (int)hipMemsetAsync(counts, 0, 8, s);
  if (err || N == 0) return err;
  int64_t blocks = (N + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(hub_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, (int)N, cap, (int)capacity, vrows, hubs, counts);
  return (int)hipGetLastError();
}
