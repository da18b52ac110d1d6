// Tall-skinny GEMM for bf16 storage (BASELINE configs[4]: bf16 activations / weights, fp32 accumulation):
//     C[N,J] = A[N,R] * Bt[J,R]^T (+ bias[J]) (+ residual[N,J]),   N >> R, J;   A, Bt, residual, C bf16
//
// Reference: the nn.Linear layers of MLP (models/gcn_lib/sparse/torch_nn.py:54-75) on every node row, forward
// (A = activations, Bt = weight) and input gradient (A = grad_out, Bt = weight^T).  bf16 operands go to
// v_mfma_f32_32x32x16_bf16 as they are (one MFMA per product, fp32 accumulate, one rounding at the store):
// 2*N*R*J FLOP at the bf16 rate is a few percent of the time the operands take to stream, so the kernel is
// HBM-bound on  N*R*2  read (once per column slice) +  N*J*2  written.
//
// Layout.  The weight is cut into column slices of JT 32-column tiles whose image (32 JT * R * 2 bytes <= 128 KB)
// stays in LDS for the whole launch in B-fragment order; blockIdx.y = slice, blockIdx.x = persistent workgroup
// that deals 32-row tiles of A round robin to its 8 waves.  Lane (r31, h) of a wave owns half of row r31 of
// the tile: one 16-byte load per k-step (k = 16 s + 8 h ..+8), eight k-steps requested ahead of their MFMAs.
// Output columns are permuted inside each PAIR of tiles -- MFMA column c of tiles (2u, 2u+1) is output column
// 64 u + 2 c + (0, 1) -- so that a lane packs its two results into one 4-byte store and a half wave writes 128
// contiguous bytes of a row (2-byte stores of the natural layout would touch 64-byte pieces).  The permutation is
// free: it only changes which weight row goes where in the LDS image.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTbBlock = 512;
constexpr int kTbWaves = kTbBlock / kWave;
constexpr int kTbMaxLds = 128 * 1024;
constexpr int kTbAhead = 8;                 // k-steps of A in flight per wave

// output column (within a slice) of MFMA column c of tile t
template <int JT>
__host__ __device__ __forceinline__ int tb_col(int t, int c) {
  if constexpr (JT == 1) return c;
  return 64 * (t >> 1) + 2 * c + (t & 1);
}

// Bt [J,R] bf16 -> image[slice][kstep][tile][lane] (16 bytes each): lane l of tile t, k-step s of slice q holds
// Bt[32 JT q + tb_col(t, l & 31)][16 s + 8 (l >> 5) .. + 8]
template <int JT>
__global__ __launch_bounds__(256) void tallgemm_bf16_pack_kernel(const uint4* __restrict__ bt, uint4* __restrict__ image,
                                                                 int J, int R) {
  const int ksteps = R / 16, slices = J / (32 * JT);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= slices * ksteps * JT * 64) return;
  const int lane = idx & 63, t = (idx >> 6) % JT, s = ((idx >> 6) / JT) % ksteps, q = (idx >> 6) / (JT * ksteps);
  const int row = 32 * JT * q + tb_col<JT>(t, lane & 31);
  image[idx] = bt[((size_t)row * R + 16 * s + 8 * (lane >> 5)) / 8];
}

struct TbArgs {
  const uint4* a; const uint4* image; const float* bias; const uint16_t* res; uint16_t* c;
  // SHIFT (input gradient of a Linear behind a softmax aggregation): gt = c * 2^(-lse) next to c, from the rounded c
  // (bitwise what the streaming pre-pass of csrc/aggregate_bwd.hip would produce); *spread raised when |lse| > kMaxLse
  const float* lse; uint16_t* gt; int* spread;
  int N; int R; int J;
};

// SHIFT: its own instantiation (no residual; the lse words of a lane's results are requested BEFORE the k-loop, like the
// residual's in the plain kernel -- as 32 dependent 8-byte loads per tile in the epilogue they cost more than the
// streaming pre-pass they replace; requested by the plain instantiation as well they spilled it: 256 registers + scratch)
template <int JT, bool SHIFT>
__global__ __launch_bounds__(kTbBlock) void tallgemm_bf16_kernel(const TbArgs p) {
  extern __shared__ uint4 wimg[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int r31 = lane & 31, h = lane >> 5;
  const int KS = p.R / 16;
  const int slice = blockIdx.y, j0 = slice * 32 * JT;

  const int n_frag = KS * JT * 64;
  const uint4* src = p.image + (size_t)slice * n_frag;
  for (int i = threadIdx.x; i < n_frag; i += kTbBlock) wimg[i] = src[i];
  __syncthreads();

  float bias[JT];
#pragma unroll
  for (int t = 0; t < JT; ++t) bias[t] = p.bias ? p.bias[j0 + tb_col<JT>(t, r31)] : 0.f;

  float worst = 0.f;
  const int n_tiles = (p.N + 31) / 32;
  const int row_u4 = p.R / 8;                                   // uint4 per row of A
  for (int tile = blockIdx.x * kTbWaves + wave; tile < n_tiles; tile += gridDim.x * kTbWaves) {
    const int row0 = tile * 32;
    const int arow = min(row0 + r31, p.N - 1);                  // rows past N re-read the last row, never stored
    const uint4* ap = p.a + (size_t)arow * row_u4 + h;          // k-step s: ap[2 s]

    // residual words of this lane's results (two bf16 per tile pair), requested before the k-loop so that they
    // arrive under the MFMAs instead of serialising load -> add -> store per row in the epilogue
    // (JT = 8 has no registers left for that -- 128 accumulators + 64 of operand ring -- and reads it in the epilogue)
    constexpr int RP = JT >= 2 ? JT / 2 : 1;
    constexpr bool kPrefetchRes = JT <= 4;
    uint32_t resw[kPrefetchRes ? RP : 1][16];
    constexpr bool kPrefetchLse = SHIFT && JT >= 2 && JT <= 4;
    float2 lsew[kPrefetchLse ? RP : 1][16];
    if constexpr (kPrefetchLse) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * h, p.N - 1);
        const size_t base = (size_t)row * p.J + j0;
#pragma unroll
        for (int u = 0; u < RP; ++u) lsew[u][r] = *reinterpret_cast<const float2*>(p.lse + base + 64 * u + 2 * r31);
      }
    }
    if (!SHIFT && kPrefetchRes && p.res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * h, p.N - 1);
        const size_t base = (size_t)row * p.J + j0;
#pragma unroll
        for (int u = 0; u < RP; ++u) {
          if constexpr (JT == 1) resw[u][r] = p.res[base + r31];
          else if constexpr (kPrefetchRes) resw[u][r] = *reinterpret_cast<const uint32_t*>(p.res + base + 64 * u + 2 * r31);
        }
      }
    }

    f32x16 acc[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    uint4 cur[kTbAhead], nxt[kTbAhead];
#pragma unroll
    for (int u = 0; u < kTbAhead; ++u) cur[u] = ap[2 * min(u, KS - 1)];
    for (int s0 = 0; s0 < KS; s0 += kTbAhead) {
#pragma unroll
      for (int u = 0; u < kTbAhead; ++u) nxt[u] = ap[2 * min(s0 + kTbAhead + u, KS - 1)];   // clamped: no branch
#pragma unroll
      for (int u = 0; u < kTbAhead; ++u) {
        if (s0 + u < KS) {                                      // KS is a multiple of kTbAhead from R = 128 on
          const bf16x8 av = __builtin_bit_cast(bf16x8, cur[u]);
          const uint4* wf = wimg + (size_t)(s0 + u) * JT * 64 + lane;
#pragma unroll
          for (int t = 0; t < JT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, wf[t * 64]), acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < kTbAhead; ++u) cur[u] = nxt[u];
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row >= p.N) continue;
      const size_t base = (size_t)row * p.J + j0;
      if constexpr (JT == 1) {
        float v = acc[0][r] + bias[0];
        if (!SHIFT && p.res) v += bf16_to_f32((uint16_t)resw[0][r]);
        const uint16_t cb = f32_to_bf16(v);
        p.c[base + r31] = cb;
        if constexpr (SHIFT) {
          const float l = p.lse[base + r31];
          p.gt[base + r31] = f32_to_bf16(bf16_to_f32(cb) * fast_exp2(-l));
          worst = fmaxf(worst, fabsf(l));
        }
      } else {
#pragma unroll
        for (int u = 0; u < JT / 2; ++u) {
          float v0 = acc[2 * u][r] + bias[2 * u], v1 = acc[2 * u + 1][r] + bias[2 * u + 1];
          const size_t at = base + 64 * u + 2 * r31;
          if (!SHIFT && p.res) {
            const uint32_t w = kPrefetchRes ? resw[kPrefetchRes ? u : 0][r] : *reinterpret_cast<const uint32_t*>(p.res + at);
            v0 += __builtin_bit_cast(float, w << 16);
            v1 += __builtin_bit_cast(float, w & 0xffff0000u);
          }
          const uint16_t c0 = f32_to_bf16(v0), c1 = f32_to_bf16(v1);
          *reinterpret_cast<uint32_t*>(p.c + at) = (uint32_t)c0 | ((uint32_t)c1 << 16);
          if constexpr (SHIFT) {
            float2 l;
            if constexpr (kPrefetchLse) l = lsew[u][r];
            else l = *reinterpret_cast<const float2*>(p.lse + at);
            const float g0 = bf16_to_f32(c0) * fast_exp2(-l.x), g1 = bf16_to_f32(c1) * fast_exp2(-l.y);
            *reinterpret_cast<uint32_t*>(p.gt + at) = (uint32_t)f32_to_bf16(g0) | ((uint32_t)f32_to_bf16(g1) << 16);
            worst = fmaxf(worst, fmaxf(fabsf(l.x), fabsf(l.y)));
          }
        }
      }
    }
  }
  if constexpr (SHIFT) {
    // (a NaN lse fails the comparison, like in softmax_shift_kernel: the NaN then travels in gt itself)
    for (int off = 1; off < kWave; off <<= 1) worst = fmaxf(worst, __shfl_xor(worst, off));
    if (lane == 0 && worst > kMaxLse) *p.spread = 1;              // plain store: every writer stores the same value
  }
}

// tiles per column slice: the widest of 8, 4, 2, 1 that divides J / 32 and keeps the slice image within LDS
int tb_tiles_per_slice(int64_t R, int64_t J) {
  if (R <= 0 || J <= 0 || R % 16 != 0 || J % 32 != 0 || R > 1024 || J > 4096) return 0;
  for (int jt = 8; jt >= 1; jt >>= 1)
    if ((J / 32) % jt == 0 && (int64_t)32 * jt * R * 2 <= kTbMaxLds) return jt;
  return 0;
}

int tallgemm_bf16(const void* a, const void* bt, const float* bias, const void* residual, void* c, void* workspace,
                  int64_t N, int64_t R, int64_t J, hipStream_t s, const float* lse, void* gt, int* spread) {
  const int jt = tb_tiles_per_slice(R, J);
  if (jt == 0) return MLGNN_E_SHAPE;
  const int ksteps = (int)(R / 16), slices = (int)(J / (32 * jt));
  const int n_img = slices * ksteps * jt * 64;
  TbArgs p;
  p.a = (const uint4*)a; p.image = (const uint4*)workspace; p.bias = bias; p.res = (const uint16_t*)residual;
  p.c = (uint16_t*)c; p.N = (int)N; p.R = (int)R; p.J = (int)J;
  p.lse = lse; p.gt = (uint16_t*)gt; p.spread = spread;
  const size_t lds = (size_t)ksteps * jt * 64 * 16;
  const int64_t tiles = (N + 31) / 32;
  int gx = (int)((tiles + kTbWaves - 1) / kTbWaves);
  const int per_slice = 256 / slices > 0 ? 256 / slices : 1;    // one workgroup per CU in total
  if (gx > per_slice) gx = per_slice;
  const dim3 pg((n_img + 255) / 256), pb(256), g(gx, slices), b(kTbBlock);
#define MLGNN_TB_CASE(JT_)                                                                                   \
  case JT_:                                                                                                  \
    hipLaunchKernelGGL((tallgemm_bf16_pack_kernel<JT_>), pg, pb, 0, s, (const uint4*)bt, (uint4*)workspace,  \
                       (int)J, (int)R);                                                                      \
    if (lse) {                                                                                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tallgemm_bf16_kernel<JT_, true>),            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kTbMaxLds);                      \
      hipLaunchKernelGGL((tallgemm_bf16_kernel<JT_, true>), g, b, lds, s, p);                                \
    } else {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tallgemm_bf16_kernel<JT_, false>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kTbMaxLds);                      \
      hipLaunchKernelGGL((tallgemm_bf16_kernel<JT_, false>), g, b, lds, s, p);                               \
    }                                                                                                        \
    break;
  switch (jt) {
    MLGNN_TB_CASE(1) MLGNN_TB_CASE(2) MLGNN_TB_CASE(4) MLGNN_TB_CASE(8)
    default: return MLGNN_E_SHAPE;
  }
#undef MLGNN_TB_CASE
  return (int)hipGetLastError();
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_tallgemm_bf16_shift_supported(int64_t N, int64_t R, int64_t J) {
  return (N > 0 && N <= INT32_MAX && tb_tiles_per_slice(R, J) > 0 && N * J * 4 < ((int64_t)1 << 40)) ? 1 : 0;
}

extern "C" int mlgnn_tallgemm_bf16_shift(const void* a, const void* bt, const float* lse, void* c, void* grad_shifted,
                                         int32_t* shift_flag, void* workspace, int64_t workspace_bytes, int64_t N,
                                         int64_t R, int64_t J, void* stream) {
  if (N == 0) return 0;
  if (!mlgnn_tallgemm_bf16_shift_supported(N, R, J)) return MLGNN_E_SHAPE;
  if (!a || !bt || !lse || !c || !grad_shifted || !shift_flag || !workspace) return MLGNN_E_NULL;
  if (workspace_bytes < R * J * 2) return MLGNN_E_WORKSPACE;
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(bt) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0 ||
      ((reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(grad_shifted)) & 3) != 0 ||
      (reinterpret_cast<uintptr_t>(lse) & 7) != 0)
    return MLGNN_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  int err = (int)hipMemsetAsync(shift_flag, 0, 16, s);
  if (err) return err;
  return tallgemm_bf16(a, bt, nullptr, nullptr, c, workspace, N, R, J, s, lse, grad_shifted, shift_flag);
}

