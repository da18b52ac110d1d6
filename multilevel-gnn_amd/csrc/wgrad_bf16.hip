// Weight + bias gradient of y = x W^T + b for bf16 storage (BASELINE configs[4]):
//     dW[M,K] = grad_out[N,M]^T x[N,K],   db[M] = column sums of grad_out;      grad_out, x bf16; dW, db fp32
//
// Reference: the autograd of the nn.Linear layers of MLP (models/gcn_lib/sparse/torch_nn.py:54-75).
//
// The reduction runs over the node rows, so BOTH MFMA operands are needed with the node index contiguous in
// each lane while HBM holds them row-major.  gfx950 transposes on the way out of LDS: row pieces go from HBM to
// LDS as they are (16-byte loads, ds_write_b128, rows padded by 64 bytes so that the four rows of a transposed
// block fall into disjoint banks) and `ds_read_b64_tr_b16` hands every lane 4 consecutive nodes of its column;
// two such reads are the 8-deep operand of v_mfma_f32_32x32x16_bf16.  No split, no scaling: bf16 products are
// exact in the fp32 accumulator.
//
// Decomposition: a workgroup (8 waves as 2 x 4 over M x K, TM x TK tiles of 32 x 32 per wave, 8 accumulator
// tiles at most) owns an output block of 64 TM x 128 TK and a slab of node rows; stages of 32 rows are double
// buffered in LDS, one workgroup barrier per stage, the global loads of stage s+2 in flight under the MFMAs of
// stage s.  blockIdx = (slab, M block, K block); per-slab fp32 partials are summed in fixed order afterwards.
// Traffic: every M block streams x once and every K block streams grad_out once -- 2 passes over one of them at
// 512 x 256 -- against 2 N M K FLOP that take a fifth of that time on the bf16 matrix cores: HBM-bound.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));

constexpr int kWbBlock = 512;
constexpr int kWbStage = 32;                  // node rows per LDS stage (two k-steps of 16)
constexpr int kWbWavesM = 2, kWbWavesK = 4;

struct WbArgs {
  const uint16_t* go; const uint16_t* x; float* ws;
  int N; int M; int K; int out_cols;          // out_cols = M K + M: one partial per slab
};

__host__ __device__ constexpr int wb_stride(int cols) { return cols * 2 + 64; }     // bytes per LDS row

// lane's byte offset inside an image for the transposed read of the 32-column tile starting at column c0, rows
// r0 .. r0+3 of its 16-lane group's block: lane 4q+p of a group addresses row q, columns 4p..4p+3; groups 0/1
// cover columns 0-15 / 16-31 of rows 0-7 (k-half 0), groups 2/3 the same columns of rows 8-15 (k-half 1)
__device__ __forceinline__ int wb_tr_offset(int lane, int stride) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  return (8 * (g >> 1) + q) * stride + (16 * (g & 1) + 4 * p) * 2;
}

template <int TM, int TK>
__global__ __launch_bounds__(kWbBlock) void linear_wgrad_bf16_kernel(const WbArgs p) {
  constexpr int MB = 32 * TM * kWbWavesM, KB = 32 * TK * kWbWavesK;       // output block
  constexpr int SA = wb_stride(MB), SB = wb_stride(KB);
  constexpr int kImage = kWbStage * (SA + SB);                              // bytes per stage
  constexpr int CA = MB / 8, CB = KB / 8, CPR = CA + CB;                    // 16-byte chunks per row
  constexpr int kChunks = kWbStage * CPR, kPerThread = (kChunks + kWbBlock - 1) / kWbBlock;
  extern __shared__ __attribute__((aligned(16))) char lds[];                // 2 * kImage bytes
  using lds_ptr = __attribute__((address_space(3))) short4_t*;

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int wm = wave / kWbWavesK, wk = wave % kWbWavesK;
  const int m0 = blockIdx.y * MB, k0 = blockIdx.z * KB;

  // slab of stages
  const int n_stage = (p.N + kWbStage - 1) / kWbStage;
  const int s_begin = (int)((int64_t)n_stage * blockIdx.x / gridDim.x);
  const int s_end = (int)((int64_t)n_stage * (blockIdx.x + 1) / gridDim.x);

  f32x16 acc[TM][TK];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) bsum[i] = 0.f;

  // staging: chunk c = threadIdx + 512 u: row c / CPR, piece c % CPR (first CA pieces from grad_out, then x)
  uint4 regs[kPerThread];
  auto fetch = [&](int stage) {
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int c = min((int)threadIdx.x + kWbBlock * u, kChunks - 1);      // (a ragged last round re-reads the last chunk)
      const int row = stage * kWbStage + c / CPR, piece = c % CPR;
      const uint16_t* src = piece < CA ? p.go + (size_t)min(row, p.N - 1) * p.M + m0 + 8 * piece
                                       : p.x + (size_t)min(row, p.N - 1) * p.K + k0 + 8 * (piece - CA);
      uint4 v = *reinterpret_cast<const uint4*>(src);
      if (row >= p.N) v = make_uint4(0, 0, 0, 0);               // rows past the end contribute zeros
      regs[u] = v;
    }
  };
  auto commit = [&](int buf) {
    char* base = lds + buf * kImage;
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int c = threadIdx.x + kWbBlock * u;
      if (c >= kChunks) break;
      const int row = c / CPR, piece = c % CPR;
      char* dst = piece < CA ? base + row * SA + 16 * piece : base + kWbStage * SA + row * SB + 16 * (piece - CA);
      *reinterpret_cast<uint4*>(dst) = regs[u];
    }
  };

  const int offa = wb_tr_offset(lane, SA) + (wm * TM * 32) * 2;
  const int offb = kWbStage * SA + wb_tr_offset(lane, SB) + (wk * TK * 32) * 2;

  if (s_begin < s_end) {
    fetch(s_begin);
    commit(0);
    if (s_begin + 1 < s_end) fetch(s_begin + 1);
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
      const int buf = (s - s_begin) & 1;
      if (s + 1 < s_end) commit(buf ^ 1);                       // stage s+1 (loaded during the previous iteration)
      if (s + 2 < s_end) fetch(s + 2);
      const char* img = lds + buf * kImage;
#pragma unroll
      for (int ks = 0; ks < kWbStage / 16; ++ks) {
        bf16x8 a[TM], b[TK];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int o = offa + ks * 16 * SA + i * 64;
          const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + o));
          const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + o + 4 * SA));
          a[i] = __builtin_bit_cast(bf16x8, short8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        }
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          const int o = offb + ks * 16 * SB + j * 64;
          const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + o));
          const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + o + 4 * SB));
          b[j] = __builtin_bit_cast(bf16x8, short8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        }
        if (wk == 0 && blockIdx.z == 0) {                        // wave-uniform: column sums of grad_out
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const uint4 w = __builtin_bit_cast(uint4, a[i]);
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              t += __builtin_bit_cast(float, ww[e] << 16) + __builtin_bit_cast(float, ww[e] & 0xffff0000u);
            bsum[i] += t;
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TK; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // partial of this slab: C/D layout col = lane & 31 (k), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (m)
  float* out = p.ws + (size_t)blockIdx.x * p.out_cols;
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int k = k0 + (wk * TK + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[(size_t)m * p.K + k] = acc[i][j][r];
      }
    }
  if (wk == 0 && blockIdx.z == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float t = bsum[i] + __shfl_xor(bsum[i], 32);         // the two k-halves of column l31
      if (half == 0) out[(size_t)p.M * p.K + m0 + (wm * TM + i) * 32 + l31] = t;
    }
  }
}

// tiles per wave along M (64 TM | M) and K (128 TK | K); 0 = shape not covered
void wb_plan(int64_t M, int64_t K, int* tm, int* tk) {
  *tm = *tk = 0;
  if (M <= 0 || K <= 0 || M % 64 != 0 || K % 128 != 0 || M > 4096 || K > 4096) return;
  *tm = M % 256 == 0 ? 4 : (M % 128 == 0 ? 2 : 1);
  *tk = K % 256 == 0 ? 2 : 1;
}

int wb_slabs(int64_t N, int64_t M, int64_t K) {
  int tm, tk;
  wb_plan(M, K, &tm, &tk);
  if (tm == 0) return 0;
  const int64_t blocks = (M / (64 * tm)) * (K / (128 * tk));
  int64_t slabs = 256 / blocks;                                   // one workgroup per CU in total
  const int64_t stages = (N + kWbStage - 1) / kWbStage;
  if (slabs > (stages + 7) / 8) slabs = (stages + 7) / 8;         // at least 8 stages per slab
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

int linear_wgrad_bf16(const void* grad_out, const void* x, float* grad_w_b, float* workspace, int64_t N, int64_t M,
                      int64_t K, hipStream_t s) {
  int tm, tk;
  wb_plan(M, K, &tm, &tk);
  if (tm == 0) return MLGNN_E_SHAPE;
  const int slabs = wb_slabs(N, M, K);
  WbArgs a;
  a.go = (const uint16_t*)grad_out; a.x = (const uint16_t*)x; a.ws = workspace;
  a.N = (int)N; a.M = (int)M; a.K = (int)K; a.out_cols = (int)(M * K + M);
  const dim3 grid(slabs, (unsigned)(M / (64 * tm)), (unsigned)(K / (128 * tk))), block(kWbBlock);
#define MLGNN_WB_CASE(TM_, TK_)                                                                               \
  if (tm == TM_ && tk == TK_) {                                                                               \
    const int lds = 2 * kWbStage * (wb_stride(64 * TM_) + wb_stride(128 * TK_));                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_wgrad_bf16_kernel<TM_, TK_>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);                               \
    hipLaunchKernelGGL((linear_wgrad_bf16_kernel<TM_, TK_>), grid, block, lds, s, a);                         \
  }
  MLGNN_WB_CASE(4, 2) MLGNN_WB_CASE(4, 1) MLGNN_WB_CASE(2, 2) MLGNN_WB_CASE(2, 1) MLGNN_WB_CASE(1, 2) MLGNN_WB_CASE(1, 1)
#undef MLGNN_WB_CASE
  int err = (int)hipGetLastError();
  if (err) return err;
  launch_reduce_partials(workspace, grad_w_b, slabs, a.out_cols, s);
  return (int)hipGetLastError();
}

}  // namespace mlgnn
