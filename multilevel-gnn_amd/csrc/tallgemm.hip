// Tall-skinny fp32 GEMM on the fp16 matrix cores with scaled 2-way split precision:
//     C[N,J] = A[N,R] * Bt[J,R]^T (+ bias[J]) (+ residual[N,J]),     N >> R, J  (R, J <= 256)
//
// Reference: the nn.Linear layers of MLP (models/gcn_lib/sparse/torch_nn.py:54-75) applied to every
// node row -- forward (A = activations, Bt = weight) and input gradient (A = grad_out, Bt = weight^T).
// The fp32 MFMA runs at 1/16 of the 16-bit rate, and a 640 000 x 128 x 256 product is MFMA-bound on it
// (~0.4 ms on the library); on the 16-bit cores the same product is HBM-bound.
//
// Numerics.  Every operand is first scaled by an exact power of two so that its largest magnitude lands
// in [2^13, 2^14) -- per ROW for A (the row maximum is known once the wave holds the row), globally for
// Bt -- and then split  x = x_hi + x_lo,  x_hi = fp16(x),  x_lo = fp16(x - x_hi).  With the scaling
// x_lo stays a normal fp16 number for every element within 2^-17 of the maximum, so
// |x - x_hi - x_lo| <= 2^-22 |x| there and <= 2^-39 max|x| below.  The product is accumulated in fp32
// from three MFMAs, a_hi b_hi + a_lo b_hi + a_hi b_lo (each 11 x 11 bit product is exact in fp32; the
// dropped a_lo b_lo is <= 2^-22 |ab|), and un-scaled by the exact powers of two: relative error per
// product <= 3 * 2^-22 = 7e-7, the size of fp32 rounding in an ordinary fp32 GEMM.  (A bf16 x 3 split
// was measured first: 4.4e-6 rms, which cost a bias gradient summed over 493 000 rows its 1e-4 parity.)
//
// Persistent workgroups (one per CU): the split weight (J*R*4 bytes, <= 128 KB) sits in LDS for the
// whole launch, already in MFMA B-fragment order; each wave streams its 32-row tile of A twice with
// 16-byte loads (lane l = row l&31, k-half l>>5): once for the row maxima, once -- from L1/L2 -- through
// the k-step loop that scales, splits and issues 3 MFMAs (v_mfma_f32_32x32x16_f16) per 32-column tile.
// HBM-bound: reads N*R*4, writes N*J*4.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTgMaxLds = 128 * 1024;      // split weight image
constexpr int kTgHeader = 4;               // f16x8 slots (64 B) in front of the image: [0] = 1 / weight scale

// power of two s with  max * s in [2^13, 2^14)  and its inverse; max = 0 or denormal -> 1
__device__ __forceinline__ void pow2_scale(float max_abs, float& s, float& inv) {
  int e = (int)((__builtin_bit_cast(uint32_t, max_abs) >> 23) & 0xff);       // biased exponent
  e = min(max(e, 20), 234);
  s = __builtin_bit_cast(float, (uint32_t)(254 + 13 - e) << 23);             // 2^(13 - (e - 127))
  inv = __builtin_bit_cast(float, (uint32_t)(e - 13) << 23);                 // 2^((e - 127) - 13)
}

// Bt [J,R] fp32 -> frag[kstep][tile][hi|lo][lane][8] fp16: lane l of tile t, k-step s holds
// Bt[32 t + (l & 31)][16 s + 8 (l >> 5) + j] * scale, j = 0..7  (B operand of v_mfma_f32_32x32x16_f16).
// Every workgroup first reduces max |Bt| over the whole (<= 128 KB, L2 resident) weight itself -- cheaper
// than a separate one-workgroup launch in front -- and workgroup 0 records 1/scale in the header.
// transposed: the operand is stored [R, J] (a Linear's own weight, used for its input gradient) and read with
// swapped indices here instead of being copied into [J, R] first.
__global__ __launch_bounds__(kBlock) void tallgemm_split_weight_kernel(const float* __restrict__ bt,
                                                                       f16x8* __restrict__ image, int J, int R,
                                                                       int transposed) {
  __shared__ float red[kWavesPerBlock];
  const int n4 = J * R / 4;                                     // R % 16 == 0
  float m = 0.f;
  for (int i = threadIdx.x; i < n4; i += kBlock) {
    const float4 q = reinterpret_cast<const float4*>(bt)[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w))));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < kWavesPerBlock; ++w) m = fmaxf(m, red[w]);
  float scale, inv;
  pow2_scale(m, scale, inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(image)[0] = inv;

  const int tiles = J / 32, ksteps = R / 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (kstep, tile, lane)
  if (idx >= ksteps * tiles * 64) return;
  const int lane = idx & 63, t = (idx >> 6) % tiles, s = (idx >> 6) / tiles;
  const int row = 32 * t + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (transposed ? bt[(size_t)(k0 + j) * J + row] : bt[(size_t)row * R + k0 + j]) * scale;
    const _Float16 h = (_Float16)v;
    hi[j] = h;
    lo[j] = (_Float16)(v - (float)h);
  }
  const size_t base = kTgHeader + ((size_t)(s * tiles + t) * 2) * 64 + lane;
  image[base] = hi;
  image[base + 64] = lo;
}

struct TgArgs {
  const float* a; const f16x8* image; const float* bias; const float* res; const float* rowmax; float* c;
  const float* gamma; const float* beta; float* rstd_out; float* rowmax_out; float ln_eps;
  const float* xhat; const float* rstd_in; float* ws;        // LN = 3
  // POST: the result rows go through a SECOND LayerNorm (+ ReLU) in the epilogue -- the norm + activation the res+
  // block puts in front of the next conv (deepergcn.py:236-241) -- written to y next to c; mean / 1 sigma to
  // mean_out / rstd_out (what that LayerNorm's backward needs)
  const float* pgamma; const float* pbeta; float* y; float* mean_out; float peps; int prelu;
  // SHIFT: res holds lse [N,J] of the softmax aggregation whose backward consumes c; y receives c * 2^(-lse)
  const int* rowptr; int* spread;
  // DUAL (SAGE update, torch_vertex.py:288-291): A = [a | a2] column blocks of two tensors (k-steps [0, ks1) from a,
  // row stride 16 ks1; the rest from a2), c = leaky_relu(A Bt^T + bias, act_slope) * row_scale[row]; max |c| per row to
  // rowmax_out, max |A row| to amax_out
  const float* a2; int ks1; float act_slope; const float* row_scale; float* amax_out;
  int N; int R; int J;
};

// k-steps of the A operand in flight ahead of the MFMAs: 4 where the registers allow it, 2 for the instantiations
// that otherwise spill (8 column tiles with the LayerNorm epilogue, the POST epilogue)
#ifdef MLGNN_TG_PF
template <int JT, int KS, int LN, bool POST> constexpr int tg_prefetch() { return MLGNN_TG_PF; }
#else
#ifndef MLGNN_TG_PF3
#define MLGNN_TG_PF3 4
#endif
#ifndef MLGNN_TG_PF1
#define MLGNN_TG_PF1 2
#endif
template <int JT, int KS, int LN, bool POST> constexpr int tg_prefetch() {
  return (LN == 3 && JT == 8) ? MLGNN_TG_PF3 : (POST ? 2 : ((LN == 1 && JT == 8) ? MLGNN_TG_PF1 : 4));
}
#endif
#ifndef MLGNN_TG_LN1_LDS
#define MLGNN_TG_LN1_LDS 0
#endif
// XT: the first k-steps of a wave's NEXT row tile are requested before the epilogue of the current one, so that the
// memory pipe does not drain at every tile boundary (the epilogue is stores + row statistics: 1-2 us without a load in
// flight from this wave; with one wave per SIMD -- LN = 3 at 8 column tiles -- from the whole SIMD)
#ifdef MLGNN_TG_XT
template <int JT, int KS, int LN, bool POST, bool SHIFT> constexpr bool tg_cross_tile() { return MLGNN_TG_XT; }
#else
template <int JT, int KS, int LN, bool POST, bool SHIFT> constexpr bool tg_cross_tile() { return LN == 3 && JT == 8; }
#endif
constexpr int kTgBlock = 512;              // 8 waves: two per SIMD share the LDS image and hide each other's loads
constexpr int kTgWaves = kTgBlock / kWave;
// LN = 3 at 8 column tiles: 128 accumulator registers + the epilogue's operands do not fit the 256 registers a wave has
// at two waves per SIMD (the compiler spilled 75 of them: 0.67 ms); one wave per SIMD has 512 and runs without spills
template <int JT, int LN> constexpr int tg_block() { return (LN == 3 && JT == 8) ? 256 : kTgBlock; }

// LN = 0: plain.  LN = 1: the result rows are layer-normalised in the epilogue -- c receives
// xhat = (v - mean) * rstd (no affine), rstd_out the per-row 1/sigma and rowmax_out max |relu(gamma xhat + beta)|,
// i.e. everything the consumer needs to apply LayerNorm's affine map + ReLU on the fly (a wave holds whole
// result rows: lane (r31, h) has column 32 t + r31 of 16 rows for every tile t).  LN = 2: the A operand is such
// a normalised activation: relu(gamma[k] a + beta[k]) is applied while it is loaded.  Together they remove the
// LayerNorm+ReLU pass between the two Linears of the MLP (torch_nn.py:54-75): the hidden activation is written
// once (normalised) and read by its consumers directly.
// LN = 3: the product is the gradient arriving at such a hidden activation, dA = dY W (the MLP's second Linear run
// backwards); the epilogue takes it through the ReLU and the LayerNorm it came out of --
//     gy = dA [gamma xhat + beta > 0],  g = gamma gy,  c = rstd (g - mean(g) - xhat mean(g xhat)),
// d gamma / d beta partials per workgroup to ws, max |c| per row to rowmax_out -- so dA [N,J] is never written and
// read back (1.3 GB per layer at config 1) and the separate LayerNorm backward pass does not exist.
// POST (with LN = 2, JT <= 4): c = A Bt^T + bias (+ residual) is written as before AND layer-normalised once more,
//     y = relu?(pgamma (c - mean) rstd + pbeta),
// the pre-conv norm + ReLU of the NEXT res+ block: its separate pass (read c, write y) becomes one extra store here.
// SHIFT (with LN = 0, JT <= 4): c is the cotangent of a softmax aggregation's output (the input gradient of the Linear
// behind it); the epilogue also writes the rescaled cotangent that aggregation's backward gathers,
//     y[i][c] = c[i][c] * 2^(-lse[i][c])      (rows without incoming edges: 0),
// with lse arriving through the residual slot, and raises *spread when some |lse| > kMaxLse -- the streaming pre-pass
// of csrc/aggregate_bwd.hip (softmax_shift_kernel: read c and lse, write y) becomes one load and one store here.
template <int JT, int KS, int LN, bool POST = false, bool SHIFT = false, bool DUAL = false>   // 32-column tiles = J / 32, 16-deep k-steps = R / 16
__global__ __launch_bounds__((tg_block<JT, LN>())) void tallgemm_kernel(const TgArgs p) {
  static_assert(!POST || (LN == 2 && JT <= 4), "POST epilogue: second GEMM of the MLP, whole rows of <= 128 columns");
  static_assert(!DUAL || (LN == 0 && !POST && !SHIFT && JT <= 4), "DUAL: plain product with the activation epilogue");
  static_assert(!SHIFT || (LN == 0 && JT <= 4 && !POST), "SHIFT epilogue: plain product, lse tile in the residual registers");
  float worst_lse = 0.f;
  constexpr int kTgBlock = tg_block<JT, LN>(), kTgWaves = kTgBlock / kWave;      // (shadow the defaults above)
  extern __shared__ f16x8 wlds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int r31 = lane & 31, h = lane >> 5;
  constexpr int R = 16 * KS;

  // the split weight, once per workgroup
  constexpr int n_frag = KS * JT * 2 * 64;
  for (int i = threadIdx.x; i < n_frag; i += kTgBlock) wlds[i] = p.image[kTgHeader + i];
  __syncthreads();
  const float inv_b = reinterpret_cast<const float*>(p.image)[0];

  float bias[JT];
#pragma unroll
  for (int t = 0; t < JT; ++t) bias[t] = (LN != 3 && p.bias) ? p.bias[32 * t + r31] : 0.f;
  // LN = 1: affine parameters of this lane's output columns; LN = 2: those of the k index, staged in LDS
  float og[LN == 1 ? JT : 1], ob[LN == 1 ? JT : 1];
  float dg[LN == 3 ? JT : 1], db[LN == 3 ? JT : 1];          // LN = 3: this lane's share of d gamma / d beta
#pragma unroll
  for (int t = 0; t < (LN == 3 ? JT : 1); ++t) { dg[t] = 0.f; db[t] = 0.f; }
  float pg[POST ? JT : 1], pb[POST ? JT : 1];                  // POST: affine parameters of this lane's output columns
  if constexpr (POST) {
#pragma unroll
    for (int t = 0; t < JT; ++t) { pg[t] = p.pgamma[32 * t + r31]; pb[t] = p.pbeta[32 * t + r31]; }
  }
  float* kg = reinterpret_cast<float*>(wlds + n_frag);         // [R] gamma then [R] beta, behind the image
  constexpr bool kLn1Lds = LN == 1 && JT == 8 && MLGNN_TG_LN1_LDS;     // gamma / beta / bias of the output columns from LDS
  if constexpr (LN == 1 && !kLn1Lds) {
#pragma unroll
    for (int t = 0; t < JT; ++t) { og[t] = p.gamma[32 * t + r31]; ob[t] = p.beta[32 * t + r31]; }
  }
  if constexpr (kLn1Lds) {
    for (int i = threadIdx.x; i < 32 * JT; i += kTgBlock) {
      kg[i] = p.gamma[i]; kg[32 * JT + i] = p.beta[i]; kg[64 * JT + i] = p.bias ? p.bias[i] : 0.f;
    }
    __syncthreads();
  }
  if constexpr (LN == 3) {             // gamma / beta of the OUTPUT columns, read from LDS in the epilogue (registers are short)
    for (int i = threadIdx.x; i < 32 * JT; i += kTgBlock) { kg[i] = p.gamma[i]; kg[32 * JT + i] = p.beta[i]; }
    __syncthreads();
  }
  if constexpr (LN == 2) {
    for (int i = threadIdx.x; i < R; i += kTgBlock) { kg[i] = p.gamma[i]; kg[R + i] = p.beta[i]; }
    __syncthreads();
  }
  auto activate = [&](float (&e)[8], int s) {                  // LN = 2: a -> relu(gamma[k] a + beta[k])
    if constexpr (LN == 2) {
      const float* g = kg + 16 * s + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = relu_keep_nan(fmaf(e[j], g[j], g[R + j]));
    }
  };

  const int n_tiles = (p.N + 31) / 32;                       // 32-row tiles, dealt round robin to the waves
  const int t_stride = gridDim.x * kTgWaves;

  // Lane l owns half of row l & 31: k = 16 s + 8 (l >> 5) + j.  Two passes over the wave's 32-row tile:
  // (1) stream the rows once for the row maxima (nothing is kept -- holding a whole 1 KB row per lane
  // spills); (2) the k-step loop re-reads them (the tile is 16-32 KB: L1 / L2 hits), scales, splits and
  // multiplies, with the next k-step's 32 bytes per lane prefetched.
  constexpr int kWant = tg_prefetch<JT, KS, LN, POST>();
  // (two waves per SIMD: at most KS / 2 -- with the whole k range in the ring the k-loop unrolls completely and the
  // compiler spills; the one-wave-per-SIMD shape has the registers for a whole tile)
  constexpr int kCap = (tg_block<JT, LN>() == 256) ? KS : KS / 2;
  constexpr int PF = KS == 1 ? 1 : (kWant <= kCap ? kWant : kCap);
  constexpr bool XT = tg_cross_tile<JT, KS, LN, POST, SHIFT>();
  float4 n0[PF], n1[PF];                                       // the ring of k-steps in flight (see the k-loop)
  float m_ahead = 0.f;
  auto request = [&](int tile_) {                              // first PF k-steps (+ the row maximum) of a row tile
    const int arow_ = min(tile_ * 32 + r31, p.N - 1);
    const float* ap_ = p.a + (size_t)arow_ * R + 8 * h;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      n0[i] = *reinterpret_cast<const float4*>(ap_ + 16 * i);
      n1[i] = *reinterpret_cast<const float4*>(ap_ + 16 * i + 4);
    }
    if (p.rowmax) m_ahead = p.rowmax[arow_];
  };
  if constexpr (XT) {
    const int first = blockIdx.x * kTgWaves + wave;
    if (first < n_tiles) request(first);
  }
  for (int tile = blockIdx.x * kTgWaves + wave; tile < n_tiles; tile += t_stride) {
    const int row0 = tile * 32;
    const int arow = min(row0 + r31, p.N - 1);                 // rows past N re-read the last row, never stored
    const float* ap = p.a + (size_t)arow * (DUAL ? 16 * p.ks1 : R) + 8 * h;
    // DUAL: k-step s of the row comes from a (s < ks1) or from a2; ap2 is biased so that ap2 + 16 s is its address
    const float* ap2 = DUAL ? p.a2 + (size_t)arow * (R - 16 * p.ks1) + 8 * h - 16 * p.ks1 : nullptr;
    auto kstep = [&](int s_) { return (DUAL && s_ >= p.ks1 ? ap2 : ap) + 16 * s_; };

    float m = 0.f;
    if (p.rowmax) {                                             // the producer of A already knows max |row|
      m = XT ? m_ahead : p.rowmax[arow];
    } else if constexpr (DUAL) {
      // (nobody hands the SAGE update its operands' row maxima: up to 8 k-steps -- 16 independent 16-byte loads per
      // lane -- are requested before the first is looked at; two at a time left this pass latency-bound: a 32-row tile
      // then cost four memory round trips before its first MFMA, 250 us for 0.63 GB at config/kirc.yaml shape)
      constexpr int CH = KS < 8 ? KS : 8;
#pragma unroll
      for (int s0 = 0; s0 < KS; s0 += CH) {
        float4 q0[CH], q1[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          q0[i] = *reinterpret_cast<const float4*>(kstep(s0 + i));
          q1[i] = *reinterpret_cast<const float4*>(kstep(s0 + i) + 4);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i)
          m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(q0[i].x), fabsf(q0[i].y)), fmaxf(fabsf(q0[i].z), fabsf(q0[i].w))),
                             fmaxf(fmaxf(fabsf(q1[i].x), fabsf(q1[i].y)), fmaxf(fabsf(q1[i].z), fabsf(q1[i].w)))));
      }
      m = fmaxf(m, __shfl_xor(m, 32));
    } else {
#pragma unroll 2
      for (int s = 0; s < KS; ++s) {
        const float4 q0 = *reinterpret_cast<const float4*>(kstep(s));
        const float4 q1 = *reinterpret_cast<const float4*>(kstep(s) + 4);
        float e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        activate(e, s);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(e[j]));
      }
      m = fmaxf(m, __shfl_xor(m, 32));                          // the other half of the row
    }
    if constexpr (DUAL) {
      if (p.amax_out && h == 0 && row0 + r31 < p.N) p.amax_out[arow] = m;
    }
    float sa, inv_a;
    pow2_scale(m, sa, inv_a);
    const float unscale = inv_a * inv_b;

    // residual tile (C/D layout: register r of lane (r31, h) is row (r & 3) + 8 (r >> 2) + 4 h, column
    // 32 t + r31), requested before the k-loop so that its loads are in flight under the MFMAs instead of
    // being serialised against the stores of the epilogue (JT <= 4: 64 more registers)
    float res[JT <= 4 ? JT : 1][16];
    if constexpr (JT <= 4) {
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * h, p.N - 1);
#pragma unroll
          for (int t = 0; t < JT; ++t) res[t][r] = p.res[(size_t)row * p.J + 32 * t + r31];
        }
      }
    }

    float my_rstd = 0.f;
    if constexpr (LN == 3) my_rstd = p.rstd_in[arow];

    f32x16 acc[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // k-loop: PF k-steps of A (32 bytes per lane each) are in flight ahead of the one being multiplied -- a k-step's 3 JT
    // MFMAs take 0.15-0.3 us, a load from HBM 1-2 us, so with one k-step ahead (the first version) every k-step waited
    // out most of a memory round trip.  The ring is unrolled PF-fold; the outer loop is not (a full unroll spills).
    // (at most KS / 2: with the whole k range in the ring the loop below unrolls completely and the compiler spills)
    if constexpr (!XT) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        n0[i] = *reinterpret_cast<const float4*>(kstep(i));
        n1[i] = *reinterpret_cast<const float4*>(kstep(i) + 4);
      }
    }
#pragma unroll 1
    for (int s0 = 0; s0 < KS; s0 += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int s = s0 + i;
        float e[8] = {n0[i].x, n0[i].y, n0[i].z, n0[i].w, n1[i].x, n1[i].y, n1[i].z, n1[i].w};
        if (s + PF < KS) {
          n0[i] = *reinterpret_cast<const float4*>(kstep(s + PF));
          n1[i] = *reinterpret_cast<const float4*>(kstep(s + PF) + 4);
        }
        activate(e, s);
        f16x8 ahi, alo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = e[j] * sa;
          const _Float16 hh = (_Float16)x;
          ahi[j] = hh;
          alo[j] = (_Float16)(x - (float)hh);
        }
        const f16x8* wf = wlds + (size_t)(s * JT) * 2 * 64 + lane;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          const f16x8 bhi = wf[t * 128];
          const f16x8 blo = wf[t * 128 + 64];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[t], 0, 0, 0);
        }
      }
    }

    // XT: every slot of the ring has been consumed -- refill it with the next tile of this wave (the last tile requests
    // itself again: cache hits, and the loop stays free of a branch the load counter would have to be flushed for)
    if constexpr (XT) request(min(tile + t_stride, n_tiles - 1));

    // LN = 3: the stored normalised activation at this lane's result positions, kXhRows result rows ahead of their use
    // (32-bit byte offsets from the uniform base: [N, J] fp32 < 4 GiB, checked on the host); the accumulators of the rows
    // already written free the registers for it as the epilogue advances
    constexpr int kXhRows = JT >= 8 ? 8 : 4;
    float xh[LN == 3 ? kXhRows : 1][JT];
    auto load_xhat = [&](float (&dst)[JT], int r) {
      const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * h, p.N - 1);
      const uint32_t off = ((uint32_t)row * (uint32_t)(32 * JT) + (uint32_t)r31) * 4u;
#pragma unroll
      for (int t = 0; t < JT; ++t)
        dst[t] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.xhat) + off + 128u * t);
    };
    if constexpr (LN == 3) {
      // (compiler barrier: the gamma / beta reads of the epilogue are invariant across tiles and would otherwise be
      // hoisted out of the tile loop into registers that the k-loop needs)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < kXhRows; ++r) load_xhat(xh[r], r);
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);
    // the row scale lives in the lane that loaded that row
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float us = __shfl(unscale, rr);
      const int row = row0 + rr;
      if constexpr (LN == 1) {
        // layer-normalise row `row` (its J values sit in acc[0..JT)[r] of the 32 lanes sharing h): two-pass
        // statistics like layernorm_act_fwd_kernel, reductions over the 32-lane half (DPP rotations + one shuffle)
        float v[JT], sum = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          v[t] = fmaf(acc[t][r], us, kLn1Lds ? kg[64 * JT + 32 * t + r31] : bias[t]);
          sum += v[t];
        }
        sum = half_sum(sum);
        const float mu = sum * (1.0f / (32 * JT));
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) { v[t] -= mu; q = fmaf(v[t], v[t], q); }
        q = half_sum(q);
        const float rs = rsqrtf(q * (1.0f / (32 * JT)) + p.ln_eps);
        float ym = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          v[t] *= rs;
          const float gam = kLn1Lds ? kg[32 * t + r31] : og[t], bet = kLn1Lds ? kg[32 * JT + 32 * t + r31] : ob[t];
          ym = fmaxf(ym, fmaxf(fmaf(v[t], gam, bet), 0.f));
        }
        ym = half_max(ym);
        if (row < p.N) {
#pragma unroll
          for (int t = 0; t < JT; ++t) p.c[(size_t)row * p.J + 32 * t + r31] = v[t];
          if (r31 == 0) { p.rstd_out[row] = rs; p.rowmax_out[row] = ym; }
        }
      } else if constexpr (LN == 3) {
        const float rs = __shfl(my_rstd, rr);
        const bool live = row < p.N;                               // rows past N shadow the last row: no contribution
        float x[JT], gg[JT], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          x[t] = xh[r % kXhRows][t];
          const float gam = kg[32 * t + r31];
          const float y = fmaf(x[t], gam, kg[32 * JT + 32 * t + r31]);
          const float gy = (live && y > 0.f) ? acc[t][r] * us : 0.f;
          dg[t] = fmaf(gy, x[t], dg[t]);
          db[t] += gy;
          gg[t] = gy * gam;
          s1 += gg[t];
          s2 = fmaf(gg[t], x[t], s2);
        }
        if (r + kXhRows < 16) load_xhat(xh[r % kXhRows], r + kXhRows);     // this group of registers is free again
        s1 = half_sum(s1) * (1.0f / (32 * JT));
        s2 = half_sum(s2) * (1.0f / (32 * JT));
        float om = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          gg[t] = rs * (gg[t] - s1 - x[t] * s2);
          om = fmaxf(om, fabsf(gg[t]));
        }
        om = half_max(om);
        if (live) {
          const uint32_t off = ((uint32_t)row * (uint32_t)(32 * JT) + (uint32_t)r31) * 4u;
#pragma unroll
          for (int t = 0; t < JT; ++t) *reinterpret_cast<float*>(reinterpret_cast<char*>(p.c) + off + 128u * t) = gg[t];
          if (r31 == 0) p.rowmax_out[row] = om;
        }
        asm volatile("" ::: "memory");                             // keep the rows apart: gamma / beta are re-read per row
      } else if constexpr (POST) {
        float v[JT], sum = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          v[t] = fmaf(acc[t][r], us, bias[t]);
          if (p.res) v[t] += res[t][r];
          sum += v[t];
        }
        if (row < p.N) {
#pragma unroll
          for (int t = 0; t < JT; ++t) p.c[(size_t)row * p.J + 32 * t + r31] = v[t];
        }
        // two-pass statistics over the row (its J values sit in the 32 lanes sharing h), like layernorm_act_fwd_kernel
        const float mu = half_sum(sum) * (1.0f / (32 * JT));
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) { v[t] -= mu; q = fmaf(v[t], v[t], q); }
        const float rs = rsqrtf(half_sum(q) * (1.0f / (32 * JT)) + p.peps);
        if (row < p.N) {
#pragma unroll
          for (int t = 0; t < JT; ++t) {
            const float yv = fmaf(v[t] * rs, pg[t], pb[t]);
            p.y[(size_t)row * p.J + 32 * t + r31] = p.prelu ? relu_keep_nan(yv) : yv;
          }
          if (r31 == 0) { p.mean_out[row] = mu; p.rstd_out[row] = rs; }
        }
      } else if constexpr (SHIFT) {
        if (row < p.N) {
          const bool live = p.rowptr[row + 1] > p.rowptr[row];   // (nodes without incoming edges are never gathered)
#pragma unroll
          for (int t = 0; t < JT; ++t) {
            const float v = fmaf(acc[t][r], us, bias[t]);
            const float l = res[t][r];
            p.c[(size_t)row * p.J + 32 * t + r31] = v;
            p.y[(size_t)row * p.J + 32 * t + r31] = live ? v * fast_exp2(-l) : 0.f;
            if (live) worst_lse = fmaxf(worst_lse, fabsf(l));
          }
        }
      } else if constexpr (DUAL) {
        // SAGE update: activation, the per-row mask (value_att_mask, multilevel_gnn.py:205-207), max |row| for the
        // consumers' operand scales
        const float sc = p.row_scale ? p.row_scale[min(row, p.N - 1)] : 1.f;
        float v[JT], om = 0.f;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          const float z = fmaf(acc[t][r], us, bias[t]);
          // (slope 0 = ReLU: relu(-inf) = 0 and relu(NaN) = NaN as in torch, not -inf * 0)
          v[t] = (z > 0.f ? z : (p.act_slope == 0.f ? (z < 0.f ? 0.f : z) : z * p.act_slope)) * sc;
          om = fmaxf(om, fabsf(v[t]));
        }
        om = half_max(om);
        if (row < p.N) {
#pragma unroll
          for (int t = 0; t < JT; ++t) p.c[(size_t)row * p.J + 32 * t + r31] = v[t];
          if (r31 == 0 && p.rowmax_out) p.rowmax_out[row] = om;
        }
      } else if (row < p.N) {
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          float v = fmaf(acc[t][r], us, bias[t]);
          if constexpr (JT <= 4) { if (p.res) v += res[t][r]; }
          p.c[(size_t)row * p.J + 32 * t + r31] = v;
        }
      }
    }
  }
  if constexpr (SHIFT) {
    // (a NaN lse fails the comparison, like in softmax_shift_kernel: the NaN then travels in y itself)
    for (int off = 1; off < kWave; off <<= 1) worst_lse = fmaxf(worst_lse, __shfl_xor(worst_lse, off));
    if (lane == 0 && worst_lse > kMaxLse) *p.spread = 1;        // plain store: every writer stores the same value
  }

  if constexpr (LN == 3) {
    // d gamma / d beta: the two half-waves hold the same columns -> waves (through the LDS the weight image no longer
    // needs) -> one [2, J] partial per workgroup, summed in a fixed order by reduce_partials
    __syncthreads();
    float* red = reinterpret_cast<float*>(wlds);                 // [wave][2][J]
#pragma unroll
    for (int t = 0; t < JT; ++t) {
      dg[t] += __shfl_xor(dg[t], 32);
      db[t] += __shfl_xor(db[t], 32);
      if (h == 0) {
        red[(wave * 2 + 0) * 32 * JT + 32 * t + r31] = dg[t];
        red[(wave * 2 + 1) * 32 * JT + 32 * t + r31] = db[t];
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * 32 * JT; idx += kTgBlock) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < kTgWaves; ++w) sum += red[w * 2 * 32 * JT + idx];
      p.ws[(size_t)blockIdx.x * 2 * 32 * JT + idx] = sum;
    }
  }
}

// bf16 storage: csrc/tallgemm_bf16.hip
int tb_tiles_per_slice(int64_t R, int64_t J);
int tallgemm_bf16(const void* a, const void* bt, const float* bias, const void* residual, void* c, void* workspace,
                  int64_t N, int64_t R, int64_t J, hipStream_t s, const float* lse = nullptr, void* gt = nullptr,
                  int* spread = nullptr);

static bool tg_dims_ok(int64_t R, int64_t J) {
  const bool j_ok = (J == 32 || J == 64 || J == 128 || J == 256);
  const bool r_ok = (R == 16 || R == 32 || R == 64 || R == 128 || R == 256);
  return j_ok && r_ok && R * J * 4 <= kTgMaxLds;
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_tallgemm_supported(int64_t N, int64_t R, int64_t J, int dtype) {
  if (N <= 0 || N > INT32_MAX) return 0;
  if (dtype == MLGNN_DTYPE_BF16) return tb_tiles_per_slice(R, J) > 0 ? 1 : 0;
  return (dtype == MLGNN_DTYPE_F32 && tg_dims_ok(R, J)) ? 1 : 0;
}

// workspace: fp32 -- header + the split weight image; bf16 -- the weight in fragment order
extern "C" int64_t mlgnn_tallgemm_workspace_bytes(int64_t R, int64_t J, int dtype) {
  if (R <= 0 || J <= 0) return MLGNN_E_SHAPE;
  if (dtype == MLGNN_DTYPE_BF16) return R * J * 2;
  return R * J * 4 + kTgHeader * 16;
}

namespace mlgnn {
struct TgPost {                 // second LayerNorm of the result rows (nullptr gamma: none)
  const float* gamma; const float* beta; float eps; int relu; float* y; float* mean; float* rstd;
};
struct TgShift {                // rescaled cotangent for a softmax aggregation's backward
  const float* lse; const int* rowptr; float* gt; int* flag;
};
}

static int tallgemm_nt_any(const void* a, const void* bt, int bt_transposed, const float* bias, const void* residual,
                           const float* row_max, int ln_mode, const float* gamma, const float* beta,
                           float ln_eps, float* rstd_out, float* row_max_out, void* c, void* workspace,
                           int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, int dtype, const TgPost* post,
                           const TgShift* shift, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (N < 0 || N > INT32_MAX) return MLGNN_E_SHAPE;
  if (N == 0) return 0;
  if (dtype == MLGNN_DTYPE_BF16) {                          // plain product (+ bias, + residual) only
    if (ln_mode != 0 || bt_transposed || post || shift) return MLGNN_E_MODE;
    if (tb_tiles_per_slice(R, J) == 0) return MLGNN_E_SHAPE;
    if (!a || !bt || !c || !workspace) return MLGNN_E_NULL;
    if (workspace_bytes < R * J * 2) return MLGNN_E_WORKSPACE;
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(bt) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0 ||
        ((reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(residual)) & 3) != 0)
      return MLGNN_E_ALIGN;
    return tallgemm_bf16(a, bt, bias, residual, c, workspace, N, R, J, (hipStream_t)stream);
  }
  if (!tg_dims_ok(R, J)) return MLGNN_E_SHAPE;
  if (!a || !bt || !c || !workspace) return MLGNN_E_NULL;
  if (residual && J > 128) return MLGNN_E_SHAPE;            // the fused residual needs its tile in registers
  if (ln_mode < 0 || ln_mode > 2) return MLGNN_E_MODE;
  if (ln_mode != 0 && (!gamma || !beta)) return MLGNN_E_NULL;
  if (ln_mode == 1 && (!rstd_out || !row_max_out || residual)) return MLGNN_E_NULL;
  if (ln_mode != 0 && (R < 64 || J < 64)) return MLGNN_E_SHAPE;      // LN modes are instantiated for 64..256 only
  if (post) {
    if (ln_mode != 2 || J > 128) return MLGNN_E_MODE;
    if (!post->gamma || !post->beta || !post->y || !post->mean || !post->rstd) return MLGNN_E_NULL;
  }
  if (shift) {
    if (ln_mode != 0 || post || residual || J > 128) return MLGNN_E_MODE;
    if (!shift->lse || !shift->rowptr || !shift->gt || !shift->flag) return MLGNN_E_NULL;
  }
  if (workspace_bytes < R * J * 4 + kTgHeader * 16) return MLGNN_E_WORKSPACE;
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(bt) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
    return MLGNN_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (shift) {
    int err0 = (int)hipMemsetAsync(shift->flag, 0, 16, s);
    if (err0) return err0;
  }
  const int n_frag_lanes = (int)(R / 16) * (int)(J / 32) * 64;
  hipLaunchKernelGGL(tallgemm_split_weight_kernel, dim3((n_frag_lanes + 255) / 256), dim3(256), 0, s,
                     (const float*)bt, (f16x8*)workspace, (int)J, (int)R, bt_transposed);
  int err = (int)hipGetLastError();
  if (err) return err;
  TgArgs p;
  p.a = (const float*)a; p.image = (const f16x8*)workspace; p.bias = bias; p.res = (const float*)residual;
  p.rowmax = row_max; p.c = (float*)c;
  p.gamma = gamma; p.beta = beta; p.rstd_out = rstd_out; p.rowmax_out = row_max_out; p.ln_eps = ln_eps;
  p.xhat = nullptr; p.rstd_in = nullptr; p.ws = nullptr;
  p.pgamma = nullptr; p.pbeta = nullptr; p.y = nullptr; p.mean_out = nullptr; p.peps = 0.f; p.prelu = 0;
  if (post) {
    p.pgamma = post->gamma; p.pbeta = post->beta; p.y = post->y; p.mean_out = post->mean; p.rstd_out = post->rstd;
    p.peps = post->eps; p.prelu = post->relu;
  }
  p.rowptr = nullptr; p.spread = nullptr;
  p.a2 = nullptr; p.ks1 = 0; p.act_slope = 1.f; p.row_scale = nullptr; p.amax_out = nullptr;
  if (shift) { p.res = shift->lse; p.y = shift->gt; p.rowptr = shift->rowptr; p.spread = shift->flag; }
  p.N = (int)N; p.R = (int)R; p.J = (int)J;
  const size_t lds = (size_t)R * J * 4 + (ln_mode == 2 ? (size_t)R * 8 : 0) + (ln_mode == 1 ? (size_t)J * 12 : 0);
  const int64_t tiles = (N + 31) / 32;
  int grid = (int)((tiles + kTgWaves - 1) / kTgWaves);
  if (grid > 256) grid = 256;                      // persistent: one workgroup per CU
  const dim3 g(grid), b(kTgBlock);
  bool launched = false;
#define MLGNN_TG_LAUNCH2(JT_, KS_, LN_, POST_, SHIFT_)                                                 \
  {                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tallgemm_kernel<JT_, KS_, LN_, POST_, SHIFT_>),  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, kTgMaxLds + 4096);          \
    hipLaunchKernelGGL((tallgemm_kernel<JT_, KS_, LN_, POST_, SHIFT_>), g, b, lds, s, p);              \
    launched = true;                                                                                  \
  }
#define MLGNN_TG_LAUNCH(JT_, KS_, LN_, POST_) MLGNN_TG_LAUNCH2(JT_, KS_, LN_, POST_, false)
#define MLGNN_TG_CASE(JT_, KS_)                                                                       \
  if (!launched && ln_mode == 0 && !shift && J == 32 * JT_ && R == 16 * KS_) MLGNN_TG_LAUNCH(JT_, KS_, 0, false)
#define MLGNN_TG_CASE_SHIFT(JT_, KS_)                                                                 \
  if (!launched && shift && J == 32 * JT_ && R == 16 * KS_) MLGNN_TG_LAUNCH2(JT_, KS_, 0, false, true)
#define MLGNN_TG_CASE_LN(JT_, KS_)                                                                    \
  if (!launched && ln_mode == 1 && J == 32 * JT_ && R == 16 * KS_) MLGNN_TG_LAUNCH(JT_, KS_, 1, false)       \
  if (!launched && ln_mode == 2 && !post && J == 32 * JT_ && R == 16 * KS_) MLGNN_TG_LAUNCH(JT_, KS_, 2, false)
#define MLGNN_TG_CASE_POST(JT_, KS_)                                                                  \
  if (!launched && ln_mode == 2 && post && J == 32 * JT_ && R == 16 * KS_) MLGNN_TG_LAUNCH(JT_, KS_, 2, true)
#define MLGNN_TG_ROW(JT_) MLGNN_TG_CASE(JT_, 1) MLGNN_TG_CASE(JT_, 2) MLGNN_TG_CASE(JT_, 4) MLGNN_TG_CASE(JT_, 8)
  MLGNN_TG_ROW(1) MLGNN_TG_ROW(2) MLGNN_TG_ROW(4) MLGNN_TG_ROW(8)
  MLGNN_TG_CASE(1, 16) MLGNN_TG_CASE(2, 16) MLGNN_TG_CASE(4, 16)          // 256 x 256 exceeds the LDS image
  MLGNN_TG_CASE_LN(2, 4) MLGNN_TG_CASE_LN(2, 8) MLGNN_TG_CASE_LN(2, 16)
  MLGNN_TG_CASE_LN(4, 4) MLGNN_TG_CASE_LN(4, 8) MLGNN_TG_CASE_LN(4, 16)
  MLGNN_TG_CASE_LN(8, 4) MLGNN_TG_CASE_LN(8, 8)
  MLGNN_TG_CASE_POST(2, 4) MLGNN_TG_CASE_POST(2, 8) MLGNN_TG_CASE_POST(2, 16)
  MLGNN_TG_CASE_POST(4, 4) MLGNN_TG_CASE_POST(4, 8) MLGNN_TG_CASE_POST(4, 16)
  MLGNN_TG_CASE_SHIFT(2, 4) MLGNN_TG_CASE_SHIFT(2, 8) MLGNN_TG_CASE_SHIFT(2, 16)
  MLGNN_TG_CASE_SHIFT(4, 4) MLGNN_TG_CASE_SHIFT(4, 8) MLGNN_TG_CASE_SHIFT(4, 16)
#undef MLGNN_TG_CASE_SHIFT
#undef MLGNN_TG_LAUNCH2
#undef MLGNN_TG_ROW
#undef MLGNN_TG_CASE_POST
#undef MLGNN_TG_CASE_LN
#undef MLGNN_TG_CASE
#undef MLGNN_TG_LAUNCH
  if (!launched) return MLGNN_E_SHAPE;
  return (int)hipGetLastError();
}

extern "C" int mlgnn_tallgemm_nt(const void* a, const void* bt, int bt_transposed, const float* bias, const void* residual,
                                 const float* row_max, int ln_mode, const float* gamma, const float* beta,
                                 float ln_eps, float* rstd_out, float* row_max_out, void* c, void* workspace,
                                 int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, int dtype,
                                 void* stream) {
  return tallgemm_nt_any(a, bt, bt_transposed, bias, residual, row_max, ln_mode, gamma, beta, ln_eps, rstd_out,
                         row_max_out, c, workspace, workspace_bytes, N, R, J, dtype, nullptr, nullptr, stream);
}

extern "C" int mlgnn_tallgemm_nt_shift_supported(int64_t N, int64_t R, int64_t J) {
  const bool ok = (J == 64 || J == 128) && (R == 64 || R == 128 || R == 256) && R * J * 4 <= kTgMaxLds;
  return (N > 0 && N <= INT32_MAX && ok) ? 1 : 0;
}

extern "C" int mlgnn_tallgemm_nt_shift(const float* a, const float* bt, int bt_transposed, const float* row_max,
                                       const float* lse, const int32_t* rowptr, float* c, float* grad_shifted,
                                       int32_t* shift_flag, void* workspace, int64_t workspace_bytes, int64_t N,
                                       int64_t R, int64_t J, void* stream) {
  if (N == 0) return 0;
  if (!mlgnn_tallgemm_nt_shift_supported(N, R, J)) return MLGNN_E_SHAPE;
  TgShift shift{lse, rowptr, grad_shifted, shift_flag};
  return tallgemm_nt_any(a, bt, bt_transposed, nullptr, nullptr, row_max, 0, nullptr, nullptr, 0.f, nullptr, nullptr, c,
                         workspace, workspace_bytes, N, R, J, MLGNN_DTYPE_F32, nullptr, &shift, stream);
}

extern "C" int mlgnn_tallgemm_lnin_postln_supported(int64_t N, int64_t R, int64_t J) {
  const bool ok = (J == 64 || J == 128) && (R == 64 || R == 128 || R == 256) && R * J * 4 <= kTgMaxLds;
  return (N > 0 && N <= INT32_MAX && ok) ? 1 : 0;
}

extern "C" int mlgnn_tallgemm_lnin_postln(const float* xhat, const float* bt, const float* bias, const float* residual,
                                          const float* row_max, const float* gamma, const float* beta,
                                          const float* post_gamma, const float* post_beta, float post_eps,
                                          int post_relu, float* c, float* y, float* post_mean, float* post_rstd,
                                          void* workspace, int64_t workspace_bytes, int64_t N, int64_t R, int64_t J,
                                          void* stream) {
  if (N == 0) return 0;
  if (!mlgnn_tallgemm_lnin_postln_supported(N, R, J)) return MLGNN_E_SHAPE;
  TgPost post{post_gamma, post_beta, post_eps, post_relu, y, post_mean, post_rstd};
  return tallgemm_nt_any(xhat, bt, 0, bias, residual, row_max, 2, gamma, beta, 0.f, nullptr, nullptr, c, workspace,
                         workspace_bytes, N, R, J, MLGNN_DTYPE_F32, &post, nullptr, stream);
}

// ---- SAGE update: leaky_relu([a | a2] Bt^T + bias) * row_scale (DUAL above) ---------------------------------------------
extern "C" int mlgnn_tallgemm_dual_supported(int64_t N, int64_t R1, int64_t R2, int64_t J) {
  const int64_t R = R1 + R2;
  // (R2 = 0: one operand, the activation epilogue only)
  const bool ok = R1 >= 16 && R2 >= 0 && R1 % 16 == 0 && R2 % 16 == 0 && (R == 32 || R == 64 || R == 128 || R == 256) &&
                  (J == 32 || J == 64 || J == 128) && R * J * 4 <= kTgMaxLds;
  return (N > 0 && N <= INT32_MAX && ok) ? 1 : 0;
}

extern "C" int mlgnn_tallgemm_dual(const float* a, const float* a2, const float* bt, const float* bias, float act_slope,
                                   const float* row_scale, float* c, float* row_max_out, float* a_row_max_out,
                                   void* workspace, int64_t workspace_bytes, int64_t N, int64_t R1, int64_t R2, int64_t J,
                                   void* stream) {
  if (N == 0) return 0;
  if (!mlgnn_tallgemm_dual_supported(N, R1, R2, J)) return MLGNN_E_SHAPE;
  const int64_t R = R1 + R2;
  if (!a || (!a2 && R2 > 0) || !bt || !c || !workspace) return MLGNN_E_NULL;
  if (R2 == 0) a2 = a;                                    // (never dereferenced: every k-step comes from a)
  if (workspace_bytes < R * J * 4 + kTgHeader * 16) return MLGNN_E_WORKSPACE;
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(a2) | reinterpret_cast<uintptr_t>(bt) |
        reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
    return MLGNN_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int n_frag_lanes = (int)(R / 16) * (int)(J / 32) * 64;
  hipLaunchKernelGGL(tallgemm_split_weight_kernel, dim3((n_frag_lanes + 255) / 256), dim3(256), 0, s, bt,
                     (f16x8*)workspace, (int)J, (int)R, 0);
  int err = (int)hipGetLastError();
  if (err) return err;
  TgArgs p;
  p.a = a; p.image = (const f16x8*)workspace; p.bias = bias; p.res = nullptr; p.rowmax = nullptr; p.c = c;
  p.gamma = nullptr; p.beta = nullptr; p.rstd_out = nullptr; p.rowmax_out = row_max_out; p.ln_eps = 0.f;
  p.xhat = nullptr; p.rstd_in = nullptr; p.ws = nullptr;
  p.pgamma = nullptr; p.pbeta = nullptr; p.y = nullptr; p.mean_out = nullptr; p.peps = 0.f; p.prelu = 0;
  p.rowptr = nullptr; p.spread = nullptr;
  p.a2 = a2; p.ks1 = (int)(R1 / 16); p.act_slope = act_slope; p.row_scale = row_scale; p.amax_out = a_row_max_out;
  p.N = (int)N; p.R = (int)R; p.J = (int)J;
  const size_t lds = (size_t)R * J * 4;
  const int64_t tiles = (N + 31) / 32;
  int grid = (int)((tiles + kTgWaves - 1) / kTgWaves);
  if (grid > 256) grid = 256;
  bool launched = false;
#define MLGNN_TG_DUAL(JT_, KS_)                                                                                  \
  if (!launched && J == 32 * JT_ && R == 16 * KS_) {                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tallgemm_kernel<JT_, KS_, 0, false, false, true>),  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, kTgMaxLds + 4096);                     \
    hipLaunchKernelGGL((tallgemm_kernel<JT_, KS_, 0, false, false, true>), dim3(grid), dim3(kTgBlock), lds, s, p); \
    launched = true;                                                                                             \
  }
  MLGNN_TG_DUAL(1, 2) MLGNN_TG_DUAL(1, 4) MLGNN_TG_DUAL(1, 8) MLGNN_TG_DUAL(1, 16)
  MLGNN_TG_DUAL(2, 2) MLGNN_TG_DUAL(2, 4) MLGNN_TG_DUAL(2, 8) MLGNN_TG_DUAL(2, 16)
  MLGNN_TG_DUAL(4, 2) MLGNN_TG_DUAL(4, 4) MLGNN_TG_DUAL(4, 8) MLGNN_TG_DUAL(4, 16)
#undef MLGNN_TG_DUAL
  if (!launched) return MLGNN_E_SHAPE;
  return (int)hipGetLastError();
}

// ---- dA = go W through ReLU + LayerNorm backward in the epilogue (LN = 3 above) --------------------------------------
namespace mlgnn {
constexpr int kTgLnBwdBlocks = 256;        // persistent workgroups = rows of the d gamma / d beta partial table
}

extern "C" int mlgnn_tallgemm_lnbwd_supported(int64_t N, int64_t R, int64_t J) {
  const bool ok = (J == 64 || J == 128 || J == 256) && (R == 64 || R == 128 || R == 256) && R * J * 4 <= kTgMaxLds;
  return (N > 0 && N <= INT32_MAX && ok) ? 1 : 0;
}

extern "C" int64_t mlgnn_tallgemm_lnbwd_workspace_bytes(int64_t R, int64_t J) {
  if (R <= 0 || J <= 0) return MLGNN_E_SHAPE;
  return R * J * 4 + kTgHeader * 16 + (int64_t)kTgLnBwdBlocks * 2 * J * 4;
}

extern "C" int mlgnn_tallgemm_lnbwd(const float* go, const float* w, int w_transposed, const float* row_max,
                                    const float* xhat, const float* rstd, const float* gamma, const float* beta,
                                    float* grad_h, float* row_max_out, float* grad_gamma_beta, void* workspace,
                                    int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, void* stream) {
  if (N < 0 || N > INT32_MAX) return MLGNN_E_SHAPE;
  if (R <= 0 || J <= 0 || !mlgnn_tallgemm_lnbwd_supported(N > 0 ? N : 1, R, J)) return MLGNN_E_SHAPE;
  if (!grad_gamma_beta) return MLGNN_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) return (int)hipMemsetAsync(grad_gamma_beta, 0, 2 * J * sizeof(float), s);
  if (!go || !w || !xhat || !rstd || !gamma || !beta || !grad_h || !row_max_out || !workspace) return MLGNN_E_NULL;
  if (workspace_bytes < mlgnn_tallgemm_lnbwd_workspace_bytes(R, J)) return MLGNN_E_WORKSPACE;
  if (((reinterpret_cast<uintptr_t>(go) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
    return MLGNN_E_ALIGN;
  const int n_frag_lanes = (int)(R / 16) * (int)(J / 32) * 64;
  hipLaunchKernelGGL(tallgemm_split_weight_kernel, dim3((n_frag_lanes + 255) / 256), dim3(256), 0, s, w,
                     (f16x8*)workspace, (int)J, (int)R, w_transposed);
  int err = (int)hipGetLastError();
  if (err) return err;
  TgArgs p;
  p.image = (const f16x8*)workspace; p.bias = nullptr; p.res = nullptr;
  p.gamma = gamma; p.beta = beta; p.rstd_out = nullptr; p.ln_eps = 0.f;
  p.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + R * J * 4 + kTgHeader * 16);
  p.a2 = nullptr; p.ks1 = 0; p.act_slope = 1.f; p.row_scale = nullptr; p.amax_out = nullptr;
  p.pgamma = nullptr; p.pbeta = nullptr; p.y = nullptr; p.mean_out = nullptr; p.peps = 0.f; p.prelu = 0;
  p.rowptr = nullptr; p.spread = nullptr;
  p.R = (int)R; p.J = (int)J;
  size_t lds = (size_t)R * J * 4 + (size_t)J * 8;                        // weight image, then gamma / beta of the J columns
  if (lds < (size_t)kTgWaves * 2 * J * 4) lds = (size_t)kTgWaves * 2 * J * 4;   // ... re-used for the partials at the end
  const int block = J == 256 ? tg_block<8, 3>() : kTgBlock, waves = block / kWave;
  // row slabs: xhat / grad_h are addressed with 32-bit byte offsets from the slab's base (dense_slab_rows); the d gamma /
  // d beta partials of a slab are added to its predecessors' in slab order
  const int64_t slab_rows = dense_slab_rows(J);
  for (int64_t r0 = 0; r0 < N; r0 += slab_rows) {
    const int64_t n = N - r0 < slab_rows ? N - r0 : slab_rows;
    p.a = go + r0 * R; p.rowmax = row_max ? row_max + r0 : nullptr; p.c = grad_h + r0 * J;
    p.rowmax_out = row_max_out + r0; p.xhat = xhat + r0 * J; p.rstd_in = rstd + r0;
    p.N = (int)n;
    const int64_t tiles = (n + 31) / 32;
    int grid = (int)((tiles + waves - 1) / waves);
    if (grid > kTgLnBwdBlocks) grid = kTgLnBwdBlocks;
    const dim3 g(grid), b(block);
    bool launched = false;
#define MLGNN_TG_LNBWD(JT_, KS_)                                                                      \
  if (!launched && J == 32 * JT_ && R == 16 * KS_) {                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tallgemm_kernel<JT_, KS_, 3>),           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, kTgMaxLds + 4096);          \
    hipLaunchKernelGGL((tallgemm_kernel<JT_, KS_, 3>), g, b, lds, s, p);                               \
    launched = true;                                                                                  \
  }
    MLGNN_TG_LNBWD(2, 4) MLGNN_TG_LNBWD(2, 8) MLGNN_TG_LNBWD(2, 16)
    MLGNN_TG_LNBWD(4, 4) MLGNN_TG_LNBWD(4, 8) MLGNN_TG_LNBWD(4, 16)
    MLGNN_TG_LNBWD(8, 4) MLGNN_TG_LNBWD(8, 8)
#undef MLGNN_TG_LNBWD
    if (!launched) return MLGNN_E_SHAPE;
    err = (int)hipGetLastError();
    if (err) return err;
    launch_reduce_partials(p.ws, grad_gamma_beta, grid, 2 * (int)J, s, r0 > 0);
  }
  return (int)hipGetLastError();
}
