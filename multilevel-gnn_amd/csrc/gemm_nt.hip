// Large bf16 GEMM for the DiffPool contraction at BASELINE configs[4] size (pooled graph of 4096 nodes, 1024
// clusters, 256 channels: S^T Z, A S, S^T (A S), S^T S -- reference: torch_geometric dense_diff_pool as called from
// DiffPoolLayer.forward, models/diff_pooling.py:59-65).  This is the one genuinely MFMA-bound piece of the path.
//
//     C[M,N] = sum_s A_s[M,K_s] * B_s[N,K_s]^T        bf16 operands, fp32 accumulation (v_mfma_f32_16x16x32_bf16)
//
// Both operands have the contraction index contiguous ("NT"): every product of the DiffPool chain is brought into
// this form by the producers of its operands (the softmax writes S and S^T, the A S product writes T and T^T).
// Several (A_s, B_s) terms may be summed into one result: the contraction simply runs over the concatenated K range,
// which is how the backward forms  dS = Z dX'^T + T (dA'^T - cI) + T2 (dA' - cI) + S (2c G)  in ONE launch.
//
// Structure (one workgroup per CU, 8 waves: on every SIMD one MFMA wave and one loader wave):
//   * 128 x 128 output tile, K-step 64.  A 4096 x 1024 result has 256 tiles: one per CU, no split needed; smaller
//     results are split along K into fp32 slabs (fixed-order reduce afterwards: bitwise reproducible, no atomics).
//   * loader waves (4) do nothing but issue LDS-DMA (global_load_lds_dwordx4: global -> LDS without registers) two
//     K-steps ahead of the MFMAs, four 32 KB stages.  A CU takes in about 70 GB/s from L2 this way (measured: 28 us for
//     the 2 MB a workgroup of the 4096 x 1024 x 4096 product streams), which at this tile size is about as long as
//     the MFMAs themselves (30 us): when the same waves issued DMA and MFMA, a full memory queue held up the MFMAs
//     behind it (48 us); with the issue on its own waves the two overlap (39 us).
//   * MFMA waves (4) own a 64 x 64 quadrant each: 4 x 4 tiles of v_mfma_f32_16x16x32_bf16, 32 MFMAs per K-step, fragments
//     in two register sets (one per 32-deep k-step: 4 A + 4 B tiles) that are refilled one MFMA group before their
//     use, across the K-step boundary.  The 16x16x32 shape measured 7-9 % faster than 32x32x16 in this kernel (same
//     process, alternating launches: 4096 x 1024 x 4096 35 vs 39 us, 4096^3 130 vs 140 us) and needs 164 instead of
//     215 VGPRs.
//   * ONE barrier per K-step joins loaders ("K-step t + 1 has landed": counted vmcnt, the later K-steps stay in
//     flight across the barrier) and MFMA waves ("K-step t - 1 is in registers, its stage may be refilled").
//   * LDS image of a stage: [128 rows][64 k] bf16 per operand, 128-byte rows, 16-byte chunks XOR-swizzled by
//     (row >> 1) & 7 -- the DMA destination is linear, so the swizzle is applied to the per-lane SOURCE address
//     and again to the fragment read address; ds_read_b128 is then conflict-free.
//   * fragment reads are inline asm (the compiler would otherwise drain every in-flight DMA before each ds_read of
//     the array the DMA writes); their waits are tied to the fragment registers by "+v" operands so that the MFMAs
//     cannot be scheduled above them, and sched_barriers keep the read / MFMA interleave as written.
//   * workgroup -> tile: XCD-aware (workgroups b and b+8 share an XCD's L2): every XCD owns a contiguous range of
//     tiles, ordered in groups of 4 tile rows so that the range is a compact block of the output.
// Measured (MI355X, random data): 4096 x 1024 x 4096 in 35 us = 980 TFLOP/s (39 % of the 2.5 PFLOP/s dense bf16 peak);
// the L2 -> LDS ingest of one CU bounds this tile size at 28 us.  With the 32x32x16 shape the MFMA stream alone ran
// at 1.58 PFLOP/s in this kernel, with its fragment reads at 1.13.  Also measured and not kept: loader waves that
// stage through registers (global_load_dwordx4 two K-steps ahead, ds_write_b128 one K-step later -- plain loads take in
// more per CU than LDS-DMA): 46 us, the LDS write traffic next to the fragment reads costs more than the loads gain;
// all eight waves doing both DMA and MFMA (48 us), a second pair of MFMA waves splitting K with an LDS reduce (42 us),
// a fifth stage / three K-steps in flight (same), s_setprio around the MFMAs (same).
#include "gemm_nt.h"
#include "mlgnn.h"

namespace mlgnn {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int kGConsumers = 4;                                 // MFMA waves (1 per SIMD)
constexpr int kGLoaders = 4;                                   // LDS-DMA waves (1 per SIMD)
constexpr int kGThreads = (kGConsumers + kGLoaders) * kWave;
constexpr int kGStageBytes = 2 * kGemmTile * kGemmBK * 2;      // 32 KB: A tile then B tile
constexpr int kGGroupRows = 4;                                 // tile rows per ordering group
constexpr int kGCtPitch = (kGemmTile + 8) * 2;                 // bytes per row of the transposed staging image

struct GemmArgs {
  GemmDesc d;
  int tiles_m, tiles_n, ktiles;
};

template <int OFF>
__device__ __forceinline__ i32x4 lds_read16_at(uint32_t addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_dma_tiles() {              // all but the N youngest tiles (8 DMAs each) have landed
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
}

// STAGES LDS stages of 32 KB; the loader waves keep STAGES - 2 K-steps in flight behind the one being multiplied
// (whose last fragment reads may still be outstanding at the barrier) and the one about to be read.
template <int STAGES>
__global__ __launch_bounds__(kGThreads) void gemm_nt_kernel(const GemmArgs p) {
  constexpr int DEPTH = STAGES - 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

  // ---- which tile / K range -----------------------------------------------------------------------------------
  const int tiles = p.tiles_m * p.tiles_n;
  const int nwg = tiles * p.d.splits;
  int id;
  {
    const int q = nwg / kXcds, r = nwg % kXcds, xcd = blockIdx.x % kXcds;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + blockIdx.x / kXcds;
  }
  const int split = id / tiles;
  id -= split * tiles;
  int tm, tn;
  {
    const int group = kGGroupRows * p.tiles_n, gid = id / group, first = gid * kGGroupRows;
    const int gsz = min(p.tiles_m - first, kGGroupRows), within = id - gid * group;
    tm = first + within % gsz;
    tn = within / gsz;
  }
  const int m0 = tm * kGemmTile, n0 = tn * kGemmTile;
  const int t_begin = (int)((int64_t)p.ktiles * split / p.d.splits);
  const int t_end = (int)((int64_t)p.ktiles * (split + 1) / p.d.splits);
  const int T = t_end - t_begin;
  const int64_t bz = blockIdx.y;                                  // problem of a grouped launch

  if (wave >= kGConsumers) {
    // ================= loader waves: nothing but LDS-DMA issue, so that a full memory queue never holds up an MFMA
    // wave lw moves pieces (1 KB = 8 rows of a stage) lw, lw + 4, ..., lw + 28 of each operand
    const int lw = wave - kGConsumers;
    const int row0 = 8 * lw + (lane >> 3);                        // + 32 i for piece i; (row >> 1) & 7 is the same for all
    const int chunk0 = (lane & 7) ^ ((row0 >> 1) & 7);
    int seg = 0, seg_left = 0;
    const uint16_t *pa, *pb;
    int64_t step_a, step_b;                                       // 32 rows
    auto enter_segment = [&](int s, int k_tile) {
      const GemmSeg& g = p.d.seg[s];
      pa = g.a + bz * g.sa + (int64_t)(m0 + row0) * g.lda + (int64_t)k_tile * kGemmBK + chunk0 * 8;
      pb = g.b + bz * g.sb + (int64_t)(n0 + row0) * g.ldb + (int64_t)k_tile * kGemmBK + chunk0 * 8;
      step_a = 32 * g.lda;
      step_b = 32 * g.ldb;
      seg_left = g.K / kGemmBK - k_tile;
    };
    {
      int t = t_begin;
      while (t >= p.d.seg[seg].K / kGemmBK) t -= p.d.seg[seg++].K / kGemmBK;
      enter_segment(seg, t);
    }
    auto issue = [&](int stage) {
      unsigned char* dst = smem + stage * kGStageBytes + lw * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pa + i * step_a), (lds_void_t*)(dst + i * 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pb + i * step_b), (lds_void_t*)(dst + 16384 + i * 4096), 16, 0, 0);
      }
      if (--seg_left == 0 && seg + 1 < p.d.nseg) {
        enter_segment(++seg, 0);
      } else {
        pa += kGemmBK;
        pb += kGemmBK;
      }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
      if (i < T) issue(i);
    int stage_in = DEPTH % STAGES;
    for (int t = 0; t < T; ++t) {
      const int ahead = min(DEPTH - 1, T - 1 - t);                // tiles issued behind tile t
      if (ahead >= 4) wait_dma_tiles<4>();
      else if (ahead == 3) wait_dma_tiles<3>();
      else if (ahead == 2) wait_dma_tiles<2>();
      else if (ahead == 1) wait_dma_tiles<1>();
      else wait_dma_tiles<0>();
      __builtin_amdgcn_s_barrier();                               // tile t may be read; tile t - 1 is in registers
      if (t + DEPTH < T) {
        issue(stage_in);                                          // into the stage that held tile t - 2
        stage_in = stage_in + 1 == STAGES ? 0 : stage_in + 1;
      }
    }
    // the workgroup barriers of the epilogue below
    if (p.d.slab) return;
    if (p.d.dot || p.d.ct) __syncthreads();
    if (p.d.dot) __syncthreads();
    if (p.d.ct) __syncthreads();
    return;
  }

  // ================= MFMA waves: one per SIMD, a 64 x 64 quadrant each ==========================================
  // v_mfma_f32_16x16x32_bf16: 4 x 4 tiles of 16 x 16 per quadrant, two 32-deep k-steps per stage.  Lane l holds
  // A[row l & 15][k = 8 (l >> 4) ..+8] of a tile: 16-byte chunk 4 ks + (l >> 4) of the row, swizzled as the DMA wrote
  // it; tiles 1..3 of an operand are 16 rows = 2048 bytes further on (the immediate offset of the read).
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  {
    const int sw16 = (l15 >> 1) & 7;
    const uint32_t a16 = lds0 + (uint32_t)(64 * wm + l15) * 128, b16 = lds0 + 16384 + (uint32_t)(64 * wn + l15) * 128;
    const uint32_t ck0 = (uint32_t)((q4 ^ sw16) << 4), ck1 = (uint32_t)(((4 + q4) ^ sw16) << 4);
    // Fragment registers: x* hold k-step 0 of a stage, y* k-step 1 (4 A tiles and 4 B tiles each).  A set is refilled
    // one MFMA group (16 MFMAs) before its use, across the K-step boundary: at most 16 LDS reads are outstanding.
    i32x4 xa[4], xb[4], ya[4], yb[4];                            // k-step 0 / k-step 1 of a stage: 4 A and 4 B tiles
#define MLGNN_READ16(SA, SB, CK, ST)                      \
  do {                                                    \
    SA[0] = lds_read16_at<0>(a16 + (CK) + (ST));          \
    SA[1] = lds_read16_at<2048>(a16 + (CK) + (ST));       \
    SA[2] = lds_read16_at<4096>(a16 + (CK) + (ST));       \
    SA[3] = lds_read16_at<6144>(a16 + (CK) + (ST));       \
    SB[0] = lds_read16_at<0>(b16 + (CK) + (ST));          \
    SB[1] = lds_read16_at<2048>(b16 + (CK) + (ST));       \
    SB[2] = lds_read16_at<4096>(b16 + (CK) + (ST));       \
    SB[3] = lds_read16_at<6144>(b16 + (CK) + (ST));       \
  } while (0)
#define MLGNN_LANDED16(SA, SB, CNT)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                    \
               : "+v"(SA[0]), "+v"(SA[1]), "+v"(SA[2]), "+v"(SA[3]), "+v"(SB[0]), "+v"(SB[1]), "+v"(SB[2]), "+v"(SB[3])::"memory")
#define MLGNN_MFMA16(SA, SB)                                                                                          \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                  \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, SA[mt]),                    \
                                                              __builtin_bit_cast(bf16x8, SB[nt]), acc[mt][nt], 0, 0, 0)
    __builtin_amdgcn_s_barrier();                                 // K-step 0 has landed
    uint32_t st = 0;
    MLGNN_READ16(xa, xb, ck0, st);
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t + 1 < T; ++t) {
      const uint32_t st_next = st + kGStageBytes == STAGES * kGStageBytes ? 0 : st + kGStageBytes;
      MLGNN_READ16(ya, yb, ck1, st);
      MLGNN_LANDED16(xa, xb, 8);
      MLGNN_MFMA16(xa, xb);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                               // K-step t + 1 has landed
      MLGNN_READ16(xa, xb, ck0, st_next);
      MLGNN_LANDED16(ya, yb, 8);
      MLGNN_MFMA16(ya, yb);
      __builtin_amdgcn_sched_barrier(0);
      st = st_next;
    }
    MLGNN_READ16(ya, yb, ck1, st);
    MLGNN_LANDED16(xa, xb, 8);
    MLGNN_MFMA16(xa, xb);
    MLGNN_LANDED16(ya, yb, 0);
    MLGNN_MFMA16(ya, yb);
#undef MLGNN_READ16
#undef MLGNN_LANDED16
#undef MLGNN_MFMA16
  }

  // accumulator register r of tile (mi, ni): row m0 + 64 wm + 16 mi + 4 (l >> 4) + r, column n0 + 64 wn + 16 ni + (l & 15)
  const int row_base = m0 + 64 * wm + 4 * q4;
  const int col_base = n0 + 64 * wn + l15;
#define MLGNN_FOR_ACC(BODY)                                                                           \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)   \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                 \
    const int row = row_base + 16 * mi + r, col = col_base + 16 * ni;                                 \
    float v = acc[mi][ni][r];                                                                         \
    BODY                                                                                              \
    acc[mi][ni][r] = v;                                                                               \
  }

  if (p.d.slab) {
    float* slab = p.d.slab + bz * p.d.s_slab + (size_t)split * p.d.M * p.d.N;
    MLGNN_FOR_ACC(slab[(size_t)row * p.d.N + col] = v;)
    return;
  }
  if (p.d.dot || p.d.ct) __syncthreads();                         // every wave has read its last fragments: LDS is free
  if (p.d.dot) {
    float part = 0.f;
    const uint16_t* dotp = p.d.dot + bz * p.d.s_dot;
    MLGNN_FOR_ACC(part += v * bf16_to_f32(dotp[(size_t)row * p.d.lddot + col]);)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    float* wsum = reinterpret_cast<float*>(smem + 40960);        // behind the transposed staging image
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) p.d.dot_partial[bz * p.d.s_part + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  }
  if (p.d.aux) {
    const float alpha = p.d.alpha_dev ? *p.d.alpha_dev : p.d.alpha;
    MLGNN_FOR_ACC(
        const size_t at = (size_t)(bz * p.d.s_aux) + (size_t)row * p.d.ldaux + col;
        v += alpha * (p.d.aux_f32 ? reinterpret_cast<const float*>(p.d.aux)[at]
                                                   : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.d.aux)[at]));)
  }
  if (p.d.c) {
    if (p.d.c_f32) {
      MLGNN_FOR_ACC(reinterpret_cast<float*>(p.d.c)[(size_t)(bz * p.d.s_c) + (size_t)row * p.d.ldc + col] = v;)
    } else {
      MLGNN_FOR_ACC(reinterpret_cast<uint16_t*>(p.d.c)[(size_t)(bz * p.d.s_c) + (size_t)row * p.d.ldc + col] = f32_to_bf16(v);)
    }
  }
  if (p.d.ct) {
    // transposed copy through LDS: image[n][m] bf16 (pitch 272 B), a lane's 4 consecutive rows = one 8-byte write
    unsigned char* img = smem;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        uint2 w;
        w.x = (uint32_t)f32_to_bf16(acc[mi][ni][0]) | ((uint32_t)f32_to_bf16(acc[mi][ni][1]) << 16);
        w.y = (uint32_t)f32_to_bf16(acc[mi][ni][2]) | ((uint32_t)f32_to_bf16(acc[mi][ni][3]) << 16);
        *reinterpret_cast<uint2*>(img + (64 * wn + 16 * ni + l15) * kGCtPitch + (64 * wm + 16 * mi + 4 * q4) * 2) = w;
      }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int n = pass * 16 + (threadIdx.x >> 4), ch = threadIdx.x & 15;
      const uint4 v = *reinterpret_cast<const uint4*>(img + n * kGCtPitch + ch * 16);
      *reinterpret_cast<uint4*>(p.d.ct + bz * p.d.s_ct + (size_t)(n0 + n) * p.d.ldct + m0 + ch * 8) = v;
    }
  }
#undef MLGNN_FOR_ACC
}

int gemm_nt_workgroups(const GemmDesc& d) { return (d.M / kGemmTile) * (d.N / kGemmTile) * (d.splits > 0 ? d.splits : 1); }

int gemm_nt_launch(const GemmDesc& d, hipStream_t s) {
  if (d.nseg < 1 || d.nseg > kGemmMaxSeg || d.M <= 0 || d.N <= 0 || d.M % kGemmTile || d.N % kGemmTile || d.splits < 1)
    return MLGNN_E_SHAPE;
  GemmArgs p;
  p.d = d;
  p.ktiles = 0;
  for (int i = 0; i < d.nseg; ++i) {
    const GemmSeg& g = d.seg[i];
    if (!g.a || !g.b) return MLGNN_E_NULL;
    if (g.K <= 0 || g.K % kGemmBK || g.lda % 8 || g.ldb % 8 || g.lda < g.K || g.ldb < g.K) return MLGNN_E_SHAPE;
    if (((uintptr_t)g.a | (uintptr_t)g.b) & 15) return MLGNN_E_ALIGN;
    p.ktiles += g.K / kGemmBK;
  }
  if (d.splits > p.ktiles) return MLGNN_E_SHAPE;
  if (d.splits > 1 && !d.slab) return MLGNN_E_NULL;
  if (!d.slab && !d.c && !d.ct && !d.dot) return MLGNN_E_NULL;
  if (d.ct && (d.ldct % 8 || ((uintptr_t)d.ct & 15))) return MLGNN_E_ALIGN;
  p.tiles_m = d.M / kGemmTile;
  p.tiles_n = d.N / kGemmTile;
  constexpr int kStages = 4;                   // 5 (three K-steps in flight, all 160 KB of LDS) measured the same
  constexpr int lds = kStages * kGStageBytes;
  static bool attr_set = false;                // idempotent: a race only repeats the call
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<kStages>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  if (d.batch > 1) {          // every problem's operands and results must keep the 16-byte alignment checked above
    for (int i = 0; i < d.nseg; ++i)
      if (d.seg[i].sa % 8 || d.seg[i].sb % 8) return MLGNN_E_ALIGN;
    if (d.s_ct % 8) return MLGNN_E_ALIGN;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<kStages>), dim3(gemm_nt_workgroups(d), d.batch > 1 ? d.batch : 1), dim3(kGThreads), lds,
                     s, p);
  return (int)hipGetLastError();
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_gemm_bf16_nt_workgroups(int64_t M, int64_t N, int splits) {
  if (M <= 0 || N <= 0 || M % kGemmTile || N % kGemmTile || splits < 1) return 0;
  return (int)((M / kGemmTile) * (N / kGemmTile) * splits);
}

extern "C" int mlgnn_gemm_bf16_nt(const void* const* a, const void* const* b, const int64_t* lda, const int64_t* ldb,
                                  const int64_t* k, int nseg, int64_t M, int64_t N, int splits, float* slab,
                                  void* c, int64_t ldc, int c_dtype, void* ct, int64_t ldct,
                                  const void* aux, int64_t ldaux, int aux_dtype, float alpha,
                                  const void* dot, int64_t lddot, float* dot_partial, void* stream) {
  if (!a || !b || !lda || !ldb || !k) return MLGNN_E_NULL;
  if (nseg < 1 || nseg > kGemmMaxSeg || M > INT32_MAX || N > INT32_MAX) return MLGNN_E_SHAPE;
  if ((dot && !dot_partial) || (!slab && !c && !ct && !dot)) return MLGNN_E_NULL;
  GemmDesc d{};
  d.nseg = nseg;
  for (int i = 0; i < nseg; ++i) {
    if (k[i] > INT32_MAX) return MLGNN_E_SHAPE;
    d.seg[i] = GemmSeg{(const uint16_t*)a[i], (const uint16_t*)b[i], lda[i], ldb[i], (int)k[i]};
  }
  d.M = (int)M; d.N = (int)N; d.splits = splits; d.slab = slab;
  d.c = c; d.ldc = ldc; d.c_f32 = c_dtype == MLGNN_DTYPE_F32;
  d.ct = (uint16_t*)ct; d.ldct = ldct;
  d.aux = aux; d.ldaux = ldaux; d.aux_f32 = aux_dtype == MLGNN_DTYPE_F32; d.alpha = alpha;
  d.dot = (const uint16_t*)dot; d.lddot = lddot; d.dot_partial = dot_partial;
  return gemm_nt_launch(d, (hipStream_t)stream);
}
