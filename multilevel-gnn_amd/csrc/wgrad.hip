// Weight / bias gradient of a Linear layer over a very tall activation matrix:
//     dW[M,K] = A^T B,   db[M] = column sums of A,   A = grad_out [N,M],  B = x [N,K],  N >> M,K
//
// Reference: autograd of nn.Linear inside MLP (models/gcn_lib/sparse/torch_nn.py:54-75, the dense
// epilogue of every conv) -- a "TN" GEMM whose reduction dimension is the 640 000 node rows and
// whose output is 128x256: the library picks a 32x32-tile kernel without split-K (~39 TFLOP/s
// measured).  Here the rows are split over the chip (one slab per workgroup) and each wave keeps its
// share of the [M,K] output in MFMA accumulators; per-slab partials are summed in a fixed order by
// reduce_partials (bitwise reproducible).
//
// Arithmetic: every fp32 operand is split exactly into three bf16 terms x = h + m + l (8 significand
// bits each, fp32's exponent range: gradients need no scaling) and the product is accumulated in fp32
// from the six leading partial products hh + hm + mh + hl + lh + mm on v_mfma_f32_32x32x16_bf16.  The
// dropped terms (ml, lm, ll) are <= 2^-24 relative, i.e. the result carries fp32-level error
// (~2e-7 relative per product, tests/test_wgrad_gpu.py) at 6/16 of the fp32-MFMA cost per FLOP -- fast
// enough that the kernel is bound by reading A and B once from HBM.  db is summed in plain fp32.
//
// F16 (the caller knows max |A| and max |B|, e.g. from the row maxima their producers emit): the scaled two-way fp16
// split of tallgemm.hip instead -- each operand times an exact power of two that puts its largest magnitude into
// [2^13, 2^14), x = x_hi + x_lo in fp16, three MFMAs (hi hi + lo hi + hi lo) on v_mfma_f32_32x32x16_f16, exact
// un-scaling of the partial.  Elements within 2^-17 of the maximum keep 22 significant bits (3 * 2^-22 = 7e-7 per
// product at worst, rms far below), smaller ones lose them gradually down to 2^-38 of the maximum; half the MFMAs and
// two LDS planes instead of three: 0.29 -> 0.21 ms at 640 000 x 128 x 256.
#include <type_traits>

#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int kTile = 32;

struct WgradArgs {
  const float* a; const float* b; float* ws;
  const float* b_gamma; const float* b_beta;      // non-NULL: the B operand is relu(b_gamma[col] * b + b_beta[col])
  const float* a_max; const float* b_max;         // F16: kMaxParts partial maxima each of max |A| and max |B| (B after
                                                  // the affine + ReLU), written by rowmax_partials_kernel
  int N; int M; int K; int out_cols;      // out_cols = M*K + M
  int row0;                               // first row of this launch
  int rows;                               // rows of this launch (a multiple of the stage when !MASKED)
  int slot0;                              // workspace slot of workgroup 0
};

// Stage = 32 rows when the operand tiles fit the LDS budget (64 KB single buffered, 144 KB double buffered),
// 16 rows for the widest layouts.
constexpr int kSlabAlign = 32;          // the fast launch covers a multiple of this many rows (both stage sizes)
constexpr int stage_rows(int W, bool db) { return db ? (W <= 384 ? 32 : 16) : (W <= 320 ? 32 : 16); }
constexpr bool lds_fits(int W, bool db) { return (db ? 12 : 6) * stage_rows(W, db) * W <= 160 * 1024; }

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// exact three-way split of 8 floats, two at a time (v_cvt_pk_bf16_f32 rounds to nearest even; the
// widening back is a shift / mask of the packed word)
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  u32x4 hw, mw, lw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 x = {v[2 * j], v[2 * j + 1]};
    const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
    const f32x2 r1 = {x[0] - __builtin_bit_cast(float, hp << 16), x[1] - __builtin_bit_cast(float, hp & 0xffff0000u)};
    const uint32_t mp = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = {r1[0] - __builtin_bit_cast(float, mp << 16), r1[1] - __builtin_bit_cast(float, mp & 0xffff0000u)};
    const uint32_t lp = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
    hw[j] = hp; mw[j] = mp; lw[j] = lp;
  }
  h = __builtin_bit_cast(bf16x8, hw); m = __builtin_bit_cast(bf16x8, mw); l = __builtin_bit_cast(bf16x8, lw);
}

// power of two s with  max * s in [2^13, 2^14)  and its inverse; max = 0 or denormal -> 1  (as in tallgemm.hip)
__device__ __forceinline__ void wg_pow2_scale(float max_abs, float& s, float& inv) {
  int e = (int)((__builtin_bit_cast(uint32_t, max_abs) >> 23) & 0xff);       // biased exponent
  e = min(max(e, 20), 234);
  s = __builtin_bit_cast(float, (uint32_t)(254 + 13 - e) << 23);
  inv = __builtin_bit_cast(float, (uint32_t)(e - 13) << 23);
}

// F16: the operands' global maxima come from the row maxima their producers wrote ([N] each): kMaxParts slices are
// reduced by a small launch in front, every wave of the main kernel folds the partials itself (no atomics, no host
// round trip, one launch instead of two library reductions)
constexpr int kMaxParts = 256;
__global__ __launch_bounds__(256) void rowmax_partials_kernel(const float* __restrict__ a_rm, const float* __restrict__ b_rm,
                                                              int n, float* __restrict__ part) {
  __shared__ float red[2][4];
  const int per = (n + kMaxParts - 1) / kMaxParts;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  float ma = 0.f, mb = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) { ma = fmaxf(ma, a_rm[i]); mb = fmaxf(mb, b_rm[i]); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, off)); mb = fmaxf(mb, __shfl_xor(mb, off)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ma; red[1][threadIdx.x >> 6] = mb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    part[kMaxParts + blockIdx.x] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}
__device__ __forceinline__ float fold_max_parts(const float* part) {          // the same value in every lane
  const int lane = threadIdx.x & (kWave - 1);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxParts / kWave; ++i) m = fmaxf(m, part[lane + i * kWave]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  return m;
}

// scaled two-way split of 8 floats: hi = fp16(x s), lo = fp16(x s - hi)
__device__ __forceinline__ void split2(const float (&v)[8], float s, f16x8& h, f16x8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * s;
    const _Float16 hh = (_Float16)x;
    h[j] = hh;
    l[j] = (_Float16)(x - (float)hh);
  }
}

// NW waves per workgroup laid out WM x WK over the output, TM x TK tiles of 32x32 per wave.  The row slab of
// the workgroup is walked in stages of 32 (16) rows.  Work item of the loader = (operand column, group of
// 8 consecutive rows): 8 dword loads (each coalesced across the lanes: consecutive lanes hold consecutive
// columns), split into the three bf16 planes in registers, one 16-byte LDS write per plane.  LDS layout
// [plane][row group][column][8 rows] is exactly the MFMA operand layout (lane l: column l & 31, rows
// 8 (l >> 5) .. +8 of a 16-row k-step), so operand reads are linear ds_read_b128 and writes linear
// ds_write_b128: no bank conflicts, no padding.  The loads of stage s+1 are issued before stage s is
// multiplied and converted after it.  DB: two LDS buffers and one barrier per stage (the 8-wave kernel runs
// one workgroup per CU, so conversion and MFMA of different waves overlap inside the workgroup); otherwise one
// buffer, two barriers, and several workgroups per CU overlap each other.
// MASKED = false: the launch covers whole stages of unpadded operands (M, K multiples of 32): loads are
// `global_load_dword v, v_offset, s[row base]` with the per-thread byte offset fixed for the whole kernel and
// one scalar row base per load, and nothing is masked.  MASKED = true: row clamps and zero fill (tile padding,
// the last N % 32 rows).
template <int NW, int WM, int WK, int TM, int TK, bool DB, bool MASKED, bool F16>
__global__ __launch_bounds__(NW * kWave) void linear_wgrad_kernel(const WgradArgs p) {
  static_assert(WM * WK == NW, "wave layout must cover the workgroup");
  constexpr int kThreads = NW * kWave;
  constexpr int MP = WM * TM * kTile, KP = WK * TK * kTile, W = MP + KP;   // padded operand widths
  constexpr int kStageRows = stage_rows(W, DB), kGroups = kStageRows / 8;
  static_assert(lds_fits(W, DB), "operand tiles exceed LDS");
  constexpr int UA = MP * kGroups, UB = KP * kGroups;                       // (column, row group) items per stage
  constexpr int PA = (UA + kThreads - 1) / kThreads, PB = (UB + kThreads - 1) / kThreads;
  constexpr int kPlane = kGroups * W;                                       // 16-byte entries per plane
  constexpr int kPlanes = F16 ? 2 : 3;
  constexpr int kBuf = kPlanes * kPlane;
  __shared__ bf16x8 tile[(DB ? 2 : 1) * kBuf];
  static_assert(sizeof(bf16x8) == 16, "operand entry is one ds_read_b128");

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int wm = wave / WK, wk = wave % WK;
  const int half = lane >> 5, l31 = lane & 31;
  const int m_base = wm * TM * kTile, k_base = wk * TK * kTile;

  f32x16 acc[TM][TK];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // slab of this workgroup: stages [s_begin, s_end) of the launch, balanced to within one stage
  const int n_stages = (p.rows + kStageRows - 1) / kStageRows;
  const int s_begin = (int)((int64_t)n_stages * blockIdx.x / gridDim.x);
  const int s_end = (int)((int64_t)n_stages * (blockIdx.x + 1) / gridDim.x);
  const int r_begin = p.row0 + s_begin * kStageRows;
  const int r_end = min(p.row0 + p.rows, p.row0 + s_end * kStageRows);

  // per-thread work items, fixed for the whole kernel
  struct Unit { uint32_t off; int lds; int col; int grp; bool live; };
  Unit ua[PA], ub[PB];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int u = threadIdx.x + q * kThreads;
    const int uc = min(u, UA - 1);
    ua[q].col = uc % MP; ua[q].grp = uc / MP;
    ua[q].live = u < UA && ua[q].col < p.M;
    ua[q].lds = ua[q].grp * W + ua[q].col;
    ua[q].off = (uint32_t)(ua[q].grp * 8 * p.M + min(ua[q].col, p.M - 1)) * 4u;
  }
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int u = threadIdx.x + q * kThreads;
    const int uc = min(u, UB - 1);
    ub[q].col = uc % KP; ub[q].grp = uc / KP;
    ub[q].live = u < UB && ub[q].col < p.K;
    ub[q].lds = ub[q].grp * W + MP + ub[q].col;
    ub[q].off = (uint32_t)(ub[q].grp * 8 * p.K + min(ub[q].col, p.K - 1)) * 4u;
  }

  float sa = 1.f, sb = 1.f, unscale = 1.f;        // F16: operand scales (exact powers of two) and their inverse product
  if constexpr (F16) {
    float ia, ib;
    wg_pow2_scale(fold_max_parts(p.a_max), sa, ia);
    wg_pow2_scale(fold_max_parts(p.b_max), sb, ib);
    unscale = ia * ib;
  }
  constexpr int kSets = DB ? 2 : 1;               // DB: two stages of loads in flight (register sets by stage parity)
  float sa_[kSets][PA][8], sb_[kSets][PB][8];      // staged loads
  float bg[PB], bb[PB];                             // affine + ReLU applied to the B operand (layer-normalised input)
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int c = min(ub[q].col, p.K - 1);
    bg[q] = p.b_gamma ? p.b_gamma[c] : 1.f;
    bb[q] = p.b_gamma ? p.b_beta[c] : 0.f;
  }
  float bsum[PA];                                   // fp32 column sums of the A columns this thread loads
#pragma unroll
  for (int q = 0; q < PA; ++q) bsum[q] = 0.f;

  // fetch only issues loads (no use of the loaded values): they stay in flight across the multiply of the
  // previous stage
  auto fetch = [&](auto set_c, int r0) {
    constexpr int S = decltype(set_c)::value;
    if constexpr (!MASKED) {
      const char* arow = reinterpret_cast<const char*>(p.a + (size_t)r0 * p.M);      // uniform
      const char* brow = reinterpret_cast<const char*>(p.b + (size_t)r0 * p.K);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const char* aj = arow + (size_t)j * p.M * 4;
        const char* bj = brow + (size_t)j * p.K * 4;
#pragma unroll
        for (int q = 0; q < PA; ++q) sa_[S][q][j] = *reinterpret_cast<const float*>(aj + ua[q].off);
#pragma unroll
        for (int q = 0; q < PB; ++q) sb_[S][q][j] = *reinterpret_cast<const float*>(bj + ub[q].off);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int q = 0; q < PA; ++q)
          sa_[S][q][j] = p.a[(size_t)min(r0 + ua[q].grp * 8 + j, p.N - 1) * p.M + min(ua[q].col, p.M - 1)];
#pragma unroll
        for (int q = 0; q < PB; ++q)
          sb_[S][q][j] = p.b[(size_t)min(r0 + ub[q].grp * 8 + j, p.N - 1) * p.K + min(ub[q].col, p.K - 1)];
      }
    }
  };
  // valid == false (a stage past the end of the slab, DB pipeline only): the stage is committed as zeros, so
  // multiplying it is a no-op and the loop body stays free of branches (see below)
  auto commit_one = [&](bf16x8* t, const Unit& un, float (&st)[8], int r0, bool valid, float* sum, float g, float b,
                        bool act, float scale) {
    if (act) {
#pragma unroll
      for (int j = 0; j < 8; ++j) st[j] = relu_keep_nan(fmaf(st[j], g, b));
    }
    if constexpr (MASKED) {
#pragma unroll
      for (int j = 0; j < 8; ++j) st[j] = (un.live && r0 + un.grp * 8 + j < r_end) ? st[j] : 0.f;
    }
    if constexpr (DB) {
#pragma unroll
      for (int j = 0; j < 8; ++j) st[j] = valid ? st[j] : 0.f;
    }
    if (sum) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += st[j];
      *sum += s;
    }
    if constexpr (F16) {
      f16x8 h, l;
      split2(st, scale, h, l);
      t[un.lds] = __builtin_bit_cast(bf16x8, h); t[kPlane + un.lds] = __builtin_bit_cast(bf16x8, l);
    } else {
      bf16x8 h, m, l;
      split3(st, h, m, l);
      t[un.lds] = h; t[kPlane + un.lds] = m; t[2 * kPlane + un.lds] = l;
    }
  };
  auto commit = [&](auto set_c, int buf, int r0, bool valid) {
    constexpr int S = decltype(set_c)::value;
    bf16x8* t = tile + buf * kBuf;
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (UA % kThreads == 0 || threadIdx.x + q * kThreads < UA) commit_one(t, ua[q], sa_[S][q], r0, valid, &bsum[q], 1.f, 0.f, false, sa);
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (UB % kThreads == 0 || threadIdx.x + q * kThreads < UB) commit_one(t, ub[q], sb_[S][q], r0, valid, nullptr, bg[q], bb[q], p.b_gamma != nullptr, sb);
  };
  auto multiply = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < kStageRows / 16; ++kk) {
      const bf16x8* g = tile + buf * kBuf + (2 * kk + half) * W;
      bf16x8 a[TM][kPlanes], b[TK][kPlanes];
#pragma unroll
      for (int pl = 0; pl < kPlanes; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][pl] = g[pl * kPlane + m_base + i * kTile + l31];
#pragma unroll
        for (int j = 0; j < TK; ++j) b[j][pl] = g[pl * kPlane + MP + k_base + j * kTile + l31];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          f32x16 c = acc[i][j];                       // small terms first
          if constexpr (F16) {
            const f16x8 ah = __builtin_bit_cast(f16x8, a[i][0]), al = __builtin_bit_cast(f16x8, a[i][1]);
            const f16x8 bh = __builtin_bit_cast(f16x8, b[j][0]), bl = __builtin_bit_cast(f16x8, b[j][1]);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
            acc[i][j] = c;
            continue;
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  };

  // DB: stage i lives in register set i & 1 and LDS buffer i & 1; the loads of stage i+2 are issued as soon as
  // stage i has been converted, i.e. two stages ahead of the MFMAs that will consume them.  The loop body has
  // no branches: a stage index past the end re-reads the last stage (cache hit) and is committed as zeros, so
  // the compiler can count the loads in flight and waits only for the older stage (`s_waitcnt vmcnt(n)` with
  // the newer stage's loads still outstanding) -- with conditional fetches it falls back to vmcnt(0), which
  // serialises load latency and MFMAs.  Single buffered: one stage ahead (the other workgroups of the CU cover
  // the rest of the latency).
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  if (r_begin < r_end) {
    if constexpr (DB) {
      const int n = (r_end - r_begin + kStageRows - 1) / kStageRows;
      auto row_of = [&](int i) { return r_begin + min(i, n - 1) * kStageRows; };
      fetch(P0{}, row_of(0));
      fetch(P1{}, row_of(1));
      commit(P0{}, 0, row_of(0), true);
      __syncthreads();
      for (int i = 0; i < n; i += 2) {
        fetch(P0{}, row_of(i + 2));
        multiply(0);                                        // stage i
        commit(P1{}, 1, row_of(i + 1), i + 1 < n);
        __syncthreads();
        fetch(P1{}, row_of(i + 3));
        multiply(1);                                        // stage i + 1 (zeros when past the end)
        commit(P0{}, 0, row_of(i + 2), i + 2 < n);
        __syncthreads();
      }
    } else {
      fetch(P0{}, r_begin);
      for (int r0 = r_begin; r0 < r_end; r0 += kStageRows) {
        commit(P0{}, 0, r0, true);
        __syncthreads();
        if (r0 + kStageRows < r_end) fetch(P0{}, r0 + kStageRows);
        multiply(0);
        __syncthreads();
      }
    }
  }

  // partial of this slab: ws[slot][m*K + k] and ws[slot][M*K + m]
  float* out = p.ws + (size_t)(p.slot0 + blockIdx.x) * p.out_cols;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int k = k_base + j * kTile + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_base + i * kTile + (r & 3) + 8 * (r >> 2) + 4 * half;    // C/D layout of 32x32 MFMA
        if (m < p.M && k < p.K) out[(size_t)m * p.K + k] = F16 ? acc[i][j][r] * unscale : acc[i][j][r];
      }
    }
  // db: fold the row groups of each A column through LDS (fixed order)
  float* red = reinterpret_cast<float*>(tile);
#pragma unroll
  for (int q = 0; q < PA; ++q)
    if (UA % kThreads == 0 || threadIdx.x + q * kThreads < UA) red[ua[q].grp * MP + ua[q].col] = bsum[q];
  __syncthreads();
  for (int m = threadIdx.x; m < p.M; m += kThreads) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) s += red[g * MP + m];
    out[(size_t)p.M * p.K + m] = s;
  }
}

// bf16 storage: csrc/wgrad_bf16.hip
int wb_slabs(int64_t N, int64_t M, int64_t K);
int linear_wgrad_bf16(const void* grad_out, const void* x, float* grad_w_b, float* workspace, int64_t N, int64_t M,
                      int64_t K, hipStream_t s);

struct WgradPlan { int nw, wm, wk, tm, tk; };

// cheapest (wave layout) x (tiles per wave) that covers tiles_m x tiles_k with <= 4 tiles (64 accumulator
// registers) per wave: 4 waves up to 16 tiles, 8 waves (double buffered, one workgroup per CU) up to 32
static bool plan_wgrad(int tiles_m, int tiles_k, WgradPlan* out) {
  static const int layouts[7][3] = {{4, 2, 2}, {4, 4, 1}, {4, 1, 4}, {8, 4, 2}, {8, 2, 4}, {8, 8, 1}, {8, 1, 8}};
  static const int shapes[6][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {1, 4}, {4, 1}};
  int best = 1 << 30;
  bool found = false;
  for (auto& lay : layouts)
    for (auto& sh : shapes) {
      const int tm = sh[0], tk = sh[1];
      if (lay[0] == 8 && tm * tk != 4) continue;                 // 8 waves only where 4 waves run out of registers
      if (lay[1] * tm < tiles_m || lay[2] * tk < tiles_k) continue;
      if (!lds_fits((lay[1] * tm + lay[2] * tk) * kTile, lay[0] == 8)) continue;
      // padded MFMA work, then operand loads per wave, then prefer the smaller workgroup
      const int cost = ((lay[1] * tm) * (lay[2] * tk) * 16 + (tm + tk)) * 2 + (lay[0] == 8);
      if (cost < best) { best = cost; *out = {lay[0], lay[1], lay[2], tm, tk}; found = true; }
    }
  return found;
}

static int wgrad_blocks(int64_t N, const WgradPlan& pl) {
  int64_t b = (N + 511) / 512;            // at least 512 rows per slab
  const int64_t cap = pl.nw == 8 ? 256 : 512;     // one 8-wave or two 4-wave workgroups per CU
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <bool MASKED, bool F16>
static bool launch_wgrad(const WgradPlan& pl, const WgradArgs& a, int nblk, hipStream_t s) {
  const dim3 grid(nblk);
  bool launched = false;
#define MLGNN_WG_CASE(NW_, WM_, WK_, TM_, TK_)                                                       \
  if (pl.nw == NW_ && pl.wm == WM_ && pl.wk == WK_ && pl.tm == TM_ && pl.tk == TK_) {                \
    hipLaunchKernelGGL((linear_wgrad_kernel<NW_, WM_, WK_, TM_, TK_, NW_ == 8, MASKED, F16>), grid,  \
                       dim3(NW_ * kWave), 0, s, a);                                                  \
    launched = true;                                                                                 \
  }
#define MLGNN_WG_LAYOUT4(WM_, WK_)                                                                   \
  MLGNN_WG_CASE(4, WM_, WK_, 1, 1) MLGNN_WG_CASE(4, WM_, WK_, 1, 2) MLGNN_WG_CASE(4, WM_, WK_, 2, 1) \
  MLGNN_WG_CASE(4, WM_, WK_, 2, 2) MLGNN_WG_CASE(4, WM_, WK_, 1, 4) MLGNN_WG_CASE(4, WM_, WK_, 4, 1)
  MLGNN_WG_LAYOUT4(2, 2) MLGNN_WG_LAYOUT4(4, 1) MLGNN_WG_LAYOUT4(1, 4)
  MLGNN_WG_CASE(8, 4, 2, 2, 2) MLGNN_WG_CASE(8, 4, 2, 1, 4) MLGNN_WG_CASE(8, 4, 2, 4, 1)
  MLGNN_WG_CASE(8, 2, 4, 2, 2) MLGNN_WG_CASE(8, 2, 4, 1, 4) MLGNN_WG_CASE(8, 2, 4, 4, 1)
  MLGNN_WG_CASE(8, 8, 1, 2, 2) MLGNN_WG_CASE(8, 8, 1, 1, 4)          // (8,1,4,1) / (1,8,1,4): 1056-wide tiles exceed LDS
  MLGNN_WG_CASE(8, 1, 8, 2, 2) MLGNN_WG_CASE(8, 1, 8, 4, 1)
#undef MLGNN_WG_LAYOUT4
#undef MLGNN_WG_CASE
  return launched;
}

}  // namespace mlgnn

using namespace mlgnn;

// one workspace slot per workgroup of the main launch + one for the masked remainder launch
extern "C" int64_t mlgnn_linear_wgrad_workspace_floats(int64_t N, int64_t M, int64_t K, int dtype) {
  if (N < 0 || M <= 0 || K <= 0) return MLGNN_E_SHAPE;
  if (dtype == MLGNN_DTYPE_BF16) {
    const int slabs = wb_slabs(N, M, K);
    return slabs > 0 ? (int64_t)slabs * (M * K + M) : (int64_t)MLGNN_E_SHAPE;
  }
  if (dtype != MLGNN_DTYPE_F32) return MLGNN_E_DTYPE;
  WgradPlan pl;
  if (!plan_wgrad((int)((M + kTile - 1) / kTile), (int)((K + kTile - 1) / kTile), &pl)) return MLGNN_E_SHAPE;
  return (int64_t)(wgrad_blocks(N, pl) + 1) * (M * K + M) + 2 * kMaxParts;      // partials per slab, then the operands' maxima
}

extern "C" int mlgnn_linear_wgrad(const void* grad_out, const void* x, const float* x_gamma, const float* x_beta,
                                  const float* grad_out_row_max, const float* x_row_max, float* grad_w_b, float* workspace,
                                  int64_t workspace_floats, int64_t N, int64_t M, int64_t K, int dtype,
                                  void* stream) {
  if (dtype == MLGNN_DTYPE_BF16) {                          // grad_out, x bf16; grad_w_b fp32
    if (N < 0 || N > INT32_MAX || M <= 0 || K <= 0) return MLGNN_E_SHAPE;
    const int slabs = wb_slabs(N, M, K);
    if (slabs <= 0) return MLGNN_E_SHAPE;
    if (x_gamma || x_beta) return MLGNN_E_MODE;
    if (!grad_w_b || !workspace) return MLGNN_E_NULL;
    if (N == 0) return (int)hipMemsetAsync(grad_w_b, 0, (size_t)(M * K + M) * sizeof(float), (hipStream_t)stream);
    if (!grad_out || !x) return MLGNN_E_NULL;
    if (workspace_floats < (int64_t)slabs * (M * K + M)) return MLGNN_E_WORKSPACE;
    if (((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(x)) & 15) != 0) return MLGNN_E_ALIGN;
    return linear_wgrad_bf16(grad_out, x, grad_w_b, workspace, N, M, K, (hipStream_t)stream);
  }
  if (dtype != MLGNN_DTYPE_F32) return MLGNN_E_DTYPE;
  if (N < 0 || M <= 0 || K <= 0 || N > INT32_MAX || M * K > (1 << 24)) return MLGNN_E_SHAPE;
  WgradPlan pl;
  if (!plan_wgrad((int)((M + kTile - 1) / kTile), (int)((K + kTile - 1) / kTile), &pl)) return MLGNN_E_SHAPE;
  if (!grad_w_b || !workspace) return MLGNN_E_NULL;
  if (N > 0 && (!grad_out || !x)) return MLGNN_E_NULL;
  const int nblk = wgrad_blocks(N, pl);
  const int cols = (int)(M * K + M);
  if (workspace_floats < (int64_t)(nblk + 1) * cols + 2 * kMaxParts) return MLGNN_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  WgradArgs a;
  a.ws = workspace;
  a.b_gamma = x_gamma; a.b_beta = x_gamma ? x_beta : nullptr;
  const bool f16 = grad_out_row_max != nullptr && x_row_max != nullptr && N > 0;     // row maxima of both: scaled fp16 split
  a.a_max = a.b_max = nullptr;
  if (f16) {
    float* part = workspace + (int64_t)(nblk + 1) * cols;
    hipLaunchKernelGGL(rowmax_partials_kernel, dim3(kMaxParts), dim3(256), 0, s, grad_out_row_max, x_row_max, (int)N, part);
    a.a_max = part; a.b_max = part + kMaxParts;
  }
  if (x_gamma && !x_beta) return MLGNN_E_NULL;
  a.M = (int)M; a.K = (int)K; a.out_cols = cols;
  const bool padded = (M % kTile != 0) || (K % kTile != 0);
  // row slabs below 4 GiB per operand (dense_slab_rows; one slab up to 4.19 M rows x 256 columns): the operand bases
  // advance, the operand scales stay those of the whole input, slab sums are added in slab order
  const int64_t slab_rows = dense_slab_rows(M > K ? M : K);
  int64_t r0 = 0;
  do {
    const int64_t n = N - r0 < slab_rows ? N - r0 : slab_rows;
    a.a = (const float*)grad_out + r0 * M; a.b = (const float*)x + r0 * K; a.N = (int)n;
    // unpadded operands: whole stages go through the unmasked kernel, the last n % 32 rows through the masked one
    const int main_rows = padded ? 0 : (int)(n / kSlabAlign * kSlabAlign);
    const int nb_main = wgrad_blocks(n, pl);
    int slots = 0;
    if (main_rows > 0) {
      a.row0 = 0; a.rows = main_rows; a.slot0 = 0;
      if (!(f16 ? launch_wgrad<false, true>(pl, a, nb_main, s) : launch_wgrad<false, false>(pl, a, nb_main, s))) return MLGNN_E_SHAPE;
      slots = nb_main;
    }
    if (main_rows < n || n == 0) {
      a.row0 = main_rows; a.rows = (int)n - main_rows; a.slot0 = slots;
      const int nb = padded ? nb_main : 1;
      if (!(f16 ? launch_wgrad<true, true>(pl, a, nb, s) : launch_wgrad<true, false>(pl, a, nb, s))) return MLGNN_E_SHAPE;
      slots += nb;
    }
    int err = (int)hipGetLastError();
    if (err) return err;
    launch_reduce_partials(workspace, grad_w_b, slots, cols, s, r0 > 0);
    r0 += n;
  } while (r0 < N);
  return (int)hipGetLastError();
}
