// Backward of a Linear layer over a tall activation matrix in ONE pass over the cotangent:
//     dX[N,K] = go[N,M] W[M,K]        (input gradient)
//     dW[M,K] = go^T x,  db[M] = column sums of go        (weight / bias gradient)
// with the epilogue the MLP of a GENConv layer needs behind dX.
//
// Reference: autograd of the two nn.Linear layers of MLP (models/gcn_lib/sparse/torch_nn.py:54-75) inside every GENConv
// (torch_vertex.py:35,72-101).  Until round 3 each Linear ran two kernels backwards, csrc/tallgemm.hip (dX, one 32-row
// tile per wave, the weight resident in LDS) and csrc/wgrad.hip (dW, row slabs staged through LDS) -- go was streamed
// twice and so was the layer input: 0.98 GB of 2.6 GB per Linear at BASELINE configs[1].  Here a workgroup stages 32
// rows of go and x through LDS ONCE and both products are taken from that stage:
//
//   * go goes to LDS as scaled fp16 hi / lo planes in the operand layout of the dW product (entry = 8 rows of one
//     column); the dX product needs the same values with the column index contiguous per lane and reads them through
//     the transposing LDS read of gfx950 (ds_read_b64_tr_b16) -- one image, read two ways.
//   * x goes to LDS as it is (fp32, LDS-DMA: no registers in between); the one wave that owns a 32-column strip reads
//     its 16 values per lane once -- in the accumulator layout of the 32x32 MFMA, which serves as the B operand of
//     dW (the contraction index of dW is the row index: any order works as long as both operands use the same) AND as
//     the x-hat operand of the LayerNorm-backward epilogue.
//   * W (the operand of dX) lives in REGISTERS: wave c owns column strip c of dX and keeps W[:, strip] as 8 k-steps of
//     hi / lo fragments (64 registers); the 32 dW tiles are dealt 4 per wave (64 accumulator registers).
//
// Arithmetic: the scaled two-way fp16 split of csrc/tallgemm.hip / wgrad.hip (operands times an exact power of two that
// puts the largest magnitude into [2^13, 2^14), x = hi + lo in fp16, hi hi + lo hi + hi lo on v_mfma_f32_32x32x16_f16,
// fp32 accumulation, exact un-scaling): 3 * 2^-22 per product for everything within 2^-16 of the operand's maximum.
// The scales are global (from the row maxima the producers of go and x emit), not per row.
//
// Epilogues.  LN (M = 128, K = 256: the MLP's second Linear): dX is the gradient arriving at the hidden activation
// relu(gamma xhat + beta); it is taken through ReLU + LayerNorm backward (x = xhat, rstd given),
//     gy = dX [gamma xhat + beta > 0],  g = gamma gy,  dH = rstd (g - mean(g) - xhat mean(g xhat)),
// with d gamma / d beta as per-workgroup partials; a row's 256 columns are spread over the 8 waves, so the two row sums
// go through LDS once per stage.  SHIFT / PLAIN (M = 256, K = 128: the first Linear): dX is written as it is; with the
// log-sum-exp of the softmax aggregation that produced x, also dX 2^(-lse) (the rescaled cotangent that aggregation's
// backward gathers, csrc/aggregate_bwd.hip).  All partial results are reduced in a fixed order: bitwise reproducible.
#include "common.h"
#include "mlgnn.h"

namespace mlgnn {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int kLbThreads = 512, kLbWaves = 8, kLbStage = 32;
constexpr int kLbParts = 256;                 // partial maxima per operand (= the most workgroups a launch has)
enum { LB_LN = 0, LB_PLAIN = 1, LB_SHIFT = 2 };

struct LbArgs {
  const float* go; const float* w; const float* x;        // [N,M], [M,K], [N,K]
  const float* go_parts; const float* x_parts;            // [kLbParts] partial maxima of |go| and of |x'| (the dW operand)
  float* dx; float* ws; float* out_parts;                 // [N,K]; per-workgroup partial slabs; [kLbParts] max |dx| (or null)
  const float* rstd; const float* gamma; const float* beta;          // LB_LN
  const float* lse; float* gt; int* spread;                          // LB_SHIFT
  int N; int slab_cols;
  int parts_max_with_old;                                            // a later row slab: out_parts = max(old, this slab's)
};

__device__ __forceinline__ void lb_pow2_scale(float max_abs, float& s, float& inv) {    // as in tallgemm.hip
  int e = (int)((__builtin_bit_cast(uint32_t, max_abs) >> 23) & 0xff);
  e = min(max(e, 20), 234);
  s = __builtin_bit_cast(float, (uint32_t)(254 + 13 - e) << 23);
  inv = __builtin_bit_cast(float, (uint32_t)(e - 13) << 23);
}
__device__ __forceinline__ float lb_fold_parts(const float* part) {           // the same value in every lane
  const int lane = threadIdx.x & (kWave - 1);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < kLbParts / kWave; ++i) m = fmaxf(m, part[lane + i * kWave]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  return m;
}
__device__ __forceinline__ void lb_split2(const float (&v)[8], float s, f16x8& h, f16x8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * s;
    const _Float16 hh = (_Float16)x;
    h[j] = hh;
    l[j] = (_Float16)(x - (float)hh);
  }
}

// LDS reads of the main loop are inline asm: the compiler treats an LDS-DMA in flight as a pending write to the whole
// shared array and would drain every DMA (s_waitcnt vmcnt(0)) in front of any ds_read it can see
template <int OFF>
__device__ __forceinline__ float lds_read_f32(uint32_t addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ i32x4 lds_read_b128(uint32_t addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ i32x2 lds_read_tr16(uint32_t addr) {
  i32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
#define LB_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")
// The results of the asm reads / loads are tied to the wait that makes them valid ("+v"): without the tie the compiler
// is free to schedule a register-only consumer (an MFMA, a convert) in front of the s_waitcnt.
template <int N>
__device__ __forceinline__ void lgkm_landed(i32x4& a, i32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_landed(i32x4& a, i32x4& b, i32x4& c, i32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_landed(i32x2& a, i32x2& b, i32x2& c, i32x2& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_landed(float (&x)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
               : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_landed(float (&x)[16]) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                 "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
               : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_landed(float (&x)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
               : "n"(N) : "memory");
}
// global load the compiler does not count: the wait is placed by hand (vm_landed), so that it names exactly the
// operations that may stay in flight behind it
template <int IMM>
__device__ __forceinline__ float glb_load_f32(const void* base, uint32_t byte_off) {       // uniform base + 32-bit lane offset
  float v;
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(byte_off), "s"(base), "n"(IMM));
  return v;
}
template <int N>
__device__ __forceinline__ void lb_wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int ROR>
__device__ __forceinline__ float lb_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROR, 0xf, 0xf, false));
}

// rows of a 32-row stage are named in the accumulator order of the 32x32 MFMA: register r of a lane in half h is row
// rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h.  Row group G = 2 kk + h (kk = 16-deep k-step of the dW product) holds
// rho(8 kk + j, h), j = 0..7: rows 16 kk + 4 h + {0..3} and + 8 + {0..3}.
__device__ __forceinline__ constexpr int lb_rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int M, int K, int EPI>
__global__ __launch_bounds__(kLbThreads) void linear_bwd_kernel(const LbArgs p) {
  constexpr int NC = K / 32, NM = kLbWaves / NC, MW = M / NM;     // column strips of dX, splits of the m range, m per wave
  static_assert(NC * NM == kLbWaves && MW == 128, "instantiated for (M, K) = (128, 256) and (256, 128)");
  static_assert(EPI != LB_LN || NM == 1, "the LayerNorm epilogue needs the whole m range in one wave");
  constexpr int KS = MW / 16;                 // k-steps of the dX product per wave
  constexpr int MT = MW / 32;                 // dW tiles per wave
  constexpr int GS = M + 4;                   // entries per row group (+4: the four groups a transposed read touches sit 16 banks apart)
  constexpr int kPlaneB = 4 * GS * 16;        // bytes per plane (hi or lo) of one stage
  constexpr int kBufB = 2 * kPlaneB;          // hi + lo
  constexpr int kXStageB = kLbStage * K * 4;  // bytes of one x stage
  constexpr int UPT = M * 4 / kLbThreads;     // go units (column, row group) per thread
  constexpr int DMA = kXStageB / 1024 / kLbWaves;      // LDS-DMA instructions per wave and stage
  constexpr int kExB = 2 * kBufB + 3 * kXStageB;       // exchange area behind the stages
  constexpr int kTotOff = 4 * M + 64;                  // LB_LN: the per-wave totals of the row sums, [8 waves][64]
  constexpr int kExFloats = 4 * M + 64 + (NM == 2 ? kLbWaves * 512 : kLbWaves * 64);
  constexpr int kRsB = kExB + kExFloats * 4;           // LB_LN: rstd of a stage's rows, [3 stages][8 waves][64] floats
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  float* ex = reinterpret_cast<float*>(smem + kExB);

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int c = wave % NC, mh = wave / NC;
  const int col = 32 * c + r31;               // this lane's column of dX / x

  // ---- slab of stages ---------------------------------------------------------------------------------------------
  const int n_stages = (p.N + kLbStage - 1) / kLbStage;
  const int s_begin = (int)((int64_t)n_stages * blockIdx.x / gridDim.x);
  const int s_end = (int)((int64_t)n_stages * (blockIdx.x + 1) / gridDim.x);
  const int n = s_end - s_begin;

  // ---- scales -------------------------------------------------------------------------------------------------------
  float sa, ia, sx, ix, sw, iw;
  lb_pow2_scale(lb_fold_parts(p.go_parts), sa, ia);
  lb_pow2_scale(lb_fold_parts(p.x_parts), sx, ix);
  {
    float m = 0.f;
    for (int i = tid; i < M * K / 4; i += kLbThreads) {
      const float4 q = reinterpret_cast<const float4*>(p.w)[i];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) ex[wave] = m;
    __syncthreads();
    m = ex[0];
#pragma unroll
    for (int i = 1; i < kLbWaves; ++i) m = fmaxf(m, ex[i]);
    lb_pow2_scale(m, sw, iw);
    __syncthreads();
  }
  // (wave-uniform values the compiler cannot see as uniform: keep them in scalar registers)
  auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  sa = uniform(sa); sx = uniform(sx); sw = uniform(sw);
  const float ux = uniform(ia * iw), uw = uniform(ia * ix);      // un-scaling of dX and of dW

  // ---- W[:, strip] as B fragments of the dX product: lane (r31, h) of k-step s holds W[m0 + 16 s + 8 h + j][col] ----
  f16x8 wh[KS], wl[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p.w[(size_t)(mh * MW + 16 * s + 8 * h + j) * K + col];
    lb_split2(v, sw, wh[s], wl[s]);
  }
  float gam = 0.f, bet = 0.f;
  if constexpr (EPI == LB_LN) { gam = p.gamma[col]; bet = p.beta[col]; }
  const bool shift = EPI != LB_SHIFT || p.gt != nullptr;      // LB_SHIFT without lse / gt: the plain epilogue

  // ---- go units of this thread: (column, row group) ---------------------------------------------------------------
  auto ucol = [&](int q) { return (tid + q * kLbThreads) % M; };
  auto ubase = [&](int q) { const int G = (tid + q * kLbThreads) / M; return 16 * (G >> 1) + 4 * (G & 1); };
  uint32_t ulds[UPT], goff[UPT];                       // LDS entry; byte offset of (first row of the group, column) in a stage
#pragma unroll
  for (int q = 0; q < UPT; ++q) {
    ulds[q] = (uint32_t)(((tid + q * kLbThreads) / M) * GS + ucol(q)) * 16u;
    goff[q] = (uint32_t)(ubase(q) * M + ucol(q)) * 4u;
  }
  float gr[UPT][8];
  float bsum[UPT];
#pragma unroll
  for (int q = 0; q < UPT; ++q) bsum[q] = 0.f;

  auto row_of = [&](int i) { return (s_begin + min(i, n - 1)) * kLbStage; };      // stages past the slab re-read its last one
  // loads with a uniform base and 32-bit lane offsets (N M 4 < 2^32, checked on the host): one address register per
  // group of 4 rows instead of a 64-bit pair per load
  auto fetch_go = [&](int r0) {
    if (r0 + kLbStage <= p.N) {
      const float* base = p.go + (size_t)r0 * M;
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        const uint32_t o0 = goff[q], o1 = goff[q] + 8u * M * 4u;
        gr[q][0] = glb_load_f32<0>(base, o0); gr[q][1] = glb_load_f32<M * 4>(base, o0);
        gr[q][2] = glb_load_f32<2 * M * 4>(base, o0); gr[q][3] = glb_load_f32<3 * M * 4>(base, o0);
        gr[q][4] = glb_load_f32<0>(base, o1); gr[q][5] = glb_load_f32<M * 4>(base, o1);
        gr[q][6] = glb_load_f32<2 * M * 4>(base, o1); gr[q][7] = glb_load_f32<3 * M * 4>(base, o1);
      }
    } else {                                            // the ragged last stage: rows past the end re-read the last row
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = min(r0 + ubase(q) + (j & 3) + 8 * (j >> 2), p.N - 1);
          gr[q][j] = glb_load_f32<0>(p.go, (uint32_t)(row * M + ucol(q)) * 4u);
        }
      }
    }
  };
  auto commit_go = [&](int buf, int r0, bool valid) {
    const bool full = r0 + kLbStage <= p.N;
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      float v[8];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool keep = valid && (full || r0 + ubase(q) + (j & 3) + 8 * (j >> 2) < p.N);
        v[j] = keep ? gr[q][j] : 0.f;
        s += v[j];
      }
      bsum[q] += s;
      f16x8 hi, lo;
      lb_split2(v, sa, hi, lo);
      unsigned char* dst = smem + buf * kBufB + ulds[q];
      *reinterpret_cast<f16x8*>(dst) = hi;
      *reinterpret_cast<f16x8*>(dst + kPlaneB) = lo;
    }
  };
  // x stage: the 32 rows are one contiguous block of x; wave w moves pieces (1 KB) w DMA .. w DMA + DMA - 1
  auto dma_x = [&](int r0, int xbuf) {
#pragma unroll
    for (int i = 0; i < DMA; ++i) {
      const int piece = wave * DMA + i;
      const int e = piece * 256 + lane * 4;                     // float index inside the stage
      const int row = min(r0 + e / K, p.N - 1);
      const float* src = p.x + (size_t)row * K + (e % K);
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + 2 * kBufB + xbuf * kXStageB + piece * 1024), 16, 0, 0);
    }
  };

  // ---- accumulators ---------------------------------------------------------------------------------------------------
  f32x16 accw[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[t][r] = 0.f;
  float dgam = 0.f, dbet = 0.f, om = 0.f, worst = 0.f;

  // per-lane LDS addresses (bytes from the start of a stage buffer)
  const uint32_t a_w = lds0 + (uint32_t)((h * GS) + mh * MW + r31) * 16u;      // dW A fragment: group 2 kk + h, column m0 + 32 t + r31
  uint32_t a_tr;                                                                 // dX A fragment through the transposing read
  {
    const int g16 = lane >> 4, b = g16 & 1, hh = g16 >> 1, q = (lane >> 2) & 3, pp = lane & 3;
    a_tr = lds0 + (uint32_t)((2 * b + (pp & 1)) * GS + mh * MW + 8 * hh + q) * 16u + 8u * (uint32_t)(pp >> 1);
  }
  const uint32_t a_x = lds0 + 2 * kBufB + (uint32_t)((4 * h) * K + col) * 4u;  // x[rho(r, h)][col] of a stage

  constexpr int E_L = EPI == LB_LN ? 1 : (EPI == LB_SHIFT ? 8 : 0);        // vector-memory operations per iteration
  constexpr int E_S = EPI == LB_SHIFT ? 16 : (EPI == LB_LN ? 16 : 8);
  constexpr int kVmPerIter = E_L + 8 * UPT + DMA + E_S;

  if (n > 0) {
    // ---- prologue: stage 0 committed, x stages 0 and 1 on their way -------------------------------------------------
    fetch_go(row_of(0));
    dma_x(row_of(0), 0);
    dma_x(row_of(1), 1);
#pragma unroll
    for (int q = 0; q < UPT; ++q) vm_landed<0>(gr[q]);
    commit_go(0, row_of(0), true);
    __syncthreads();

    for (int i = 0; i < n; ++i) {
      const int r0 = row_of(i);
      const bool full = r0 + kLbStage <= p.N;
      const int buf = i & 1, xbuf = i % 3;
      // -- epilogue operands of THIS stage first (they are used last), then the loads of the stages ahead
      float lse8[EPI == LB_SHIFT ? 8 : 1];
      if constexpr (EPI == LB_LN) {
        // 1 / sigma of the stage's 32 rows: every wave brings its own copy into LDS (no registers held across the
        // products, and the same number of vector-memory operations in every wave)
        // (16-byte pieces from 8 lanes: the 4-byte form of the LDS-DMA does not place one dword per lane.  The last
        // piece of a ragged stage may read up to 12 bytes past rstd's end, inside its last 16-byte granule.)
        if (lane < 8)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(p.rstd + min(r0 + 4 * lane, (p.N - 1) & ~3)),
                                           (lds_void_t*)(smem + kRsB + ((i % 3) * kLbWaves + wave) * 256), 16, 0, 0);
      }
      if constexpr (EPI == LB_SHIFT) {
        if (!shift) {
          // plain epilogue through the same instantiation (a separate one without these loads makes the register
          // allocator spill the weight fragments inside the loop): eight loads of one cached word instead
#pragma unroll
          for (int q = 0; q < 8; ++q) lse8[q] = glb_load_f32<0>(p.go_parts, 0u);
        } else if (full) {
          const float* base = p.lse + (size_t)r0 * K;
          const uint32_t o0 = (uint32_t)((16 * mh + 4 * h) * K + col) * 4u, o1 = o0 + 8u * K * 4u;
          lse8[0] = glb_load_f32<0>(base, o0); lse8[1] = glb_load_f32<K * 4>(base, o0);
          lse8[2] = glb_load_f32<2 * K * 4>(base, o0); lse8[3] = glb_load_f32<3 * K * 4>(base, o0);
          lse8[4] = glb_load_f32<0>(base, o1); lse8[5] = glb_load_f32<K * 4>(base, o1);
          lse8[6] = glb_load_f32<2 * K * 4>(base, o1); lse8[7] = glb_load_f32<3 * K * 4>(base, o1);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            lse8[q] = glb_load_f32<0>(p.lse, (uint32_t)(min(r0 + lb_rho(8 * mh + q, h), p.N - 1) * K + col) * 4u);
        }
      }
      fetch_go(row_of(i + 1));
      dma_x(row_of(i + 2), (i + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);

      // -- x values of this lane: x[r0 + rho(r, h)][col], r = 0..15 (accumulator order)
      float xv[16];
      {
        const uint32_t ax = a_x + (uint32_t)xbuf * kXStageB;
#define LB_XV(R) xv[R] = lds_read_f32<(((R) & 3) + 8 * ((R) >> 2)) * K * 4>(ax);
        LB_XV(0) LB_XV(1) LB_XV(2) LB_XV(3) LB_XV(4) LB_XV(5) LB_XV(6) LB_XV(7)
        LB_XV(8) LB_XV(9) LB_XV(10) LB_XV(11) LB_XV(12) LB_XV(13) LB_XV(14) LB_XV(15)
#undef LB_XV
      }
      // -- dW += go^T x': per 16-deep k-step the B fragment from xv, the A fragments tile by tile from the planes
      const uint32_t aw = a_w + (uint32_t)buf * kBufB;
      i32x4 fa[2][2];                           // [slot][hi | lo]: the fragment pair one tile ahead
      fa[0][0] = lds_read_b128<0>(aw);
      fa[0][1] = lds_read_b128<kPlaneB>(aw);
      lgkm_landed<2>(xv);                       // (the two fragment reads may still be out)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = xv[8 * kk + j];
          if constexpr (EPI == LB_LN) v[j] = relu_keep_nan(fmaf(v[j], gam, bet));
        }
        f16x8 bh, bl;
        lb_split2(v, sx, bh, bl);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int step = kk * MT + t, slot = step & 1;
          if (step + 1 < 2 * MT) {                      // next tile's pair: group 2 kk' + h is 2 GS entries further per k-step
            constexpr int kStepB = 2 * GS * 16;
            const int nkk = (step + 1) / MT, nt = (step + 1) % MT;
            if (nkk == 0) {
              if (nt == 1) { fa[1][0] = lds_read_b128<512>(aw); fa[1][1] = lds_read_b128<kPlaneB + 512>(aw); }
              if (nt == 2) { fa[0][0] = lds_read_b128<1024>(aw); fa[0][1] = lds_read_b128<kPlaneB + 1024>(aw); }
              if (nt == 3) { fa[1][0] = lds_read_b128<1536>(aw); fa[1][1] = lds_read_b128<kPlaneB + 1536>(aw); }
            } else {
              if (nt == 0) { fa[0][0] = lds_read_b128<kStepB>(aw); fa[0][1] = lds_read_b128<kPlaneB + kStepB>(aw); }
              if (nt == 1) { fa[1][0] = lds_read_b128<kStepB + 512>(aw); fa[1][1] = lds_read_b128<kPlaneB + kStepB + 512>(aw); }
              if (nt == 2) { fa[0][0] = lds_read_b128<kStepB + 1024>(aw); fa[0][1] = lds_read_b128<kPlaneB + kStepB + 1024>(aw); }
              if (nt == 3) { fa[1][0] = lds_read_b128<kStepB + 1536>(aw); fa[1][1] = lds_read_b128<kPlaneB + kStepB + 1536>(aw); }
            }
            lgkm_landed<2>(fa[slot][0], fa[slot][1]);
          } else {
            lgkm_landed<0>(fa[slot][0], fa[slot][1]);
          }
          const f16x8 ah = __builtin_bit_cast(f16x8, fa[slot][0]), al = __builtin_bit_cast(f16x8, fa[slot][1]);
          f32x16 cw = accw[t];
          cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cw, 0, 0, 0);
          cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cw, 0, 0, 0);
          cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, cw, 0, 0, 0);
          accw[t] = cw;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // -- dX tile = go[32 rows, m range] W[m range, strip]: A fragments through the transposing read (lane = row)
      f32x16 accx;
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[r] = 0.f;
      {
        const uint32_t at = a_tr + (uint32_t)buf * kBufB;
        i32x2 ft[2][4];                          // [slot][hi m0..3, hi m4..7, lo m0..3, lo m4..7]
#define LB_TR(SLOT, S)                                                                              \
        ft[SLOT][0] = lds_read_tr16<(S) * 256>(at); ft[SLOT][1] = lds_read_tr16<(S) * 256 + 64>(at); \
        ft[SLOT][2] = lds_read_tr16<kPlaneB + (S) * 256>(at); ft[SLOT][3] = lds_read_tr16<kPlaneB + (S) * 256 + 64>(at);
#define LB_DX(SLOT, S)                                                                              \
        {                                                                                           \
          const i32x4 hi4 = {ft[SLOT][0][0], ft[SLOT][0][1], ft[SLOT][1][0], ft[SLOT][1][1]};       \
          const i32x4 lo4 = {ft[SLOT][2][0], ft[SLOT][2][1], ft[SLOT][3][0], ft[SLOT][3][1]};       \
          const f16x8 ah = __builtin_bit_cast(f16x8, hi4), al = __builtin_bit_cast(f16x8, lo4);     \
          accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[S], accx, 0, 0, 0);                  \
          accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[S], accx, 0, 0, 0);                  \
          accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[S], accx, 0, 0, 0);                  \
        }
#define LB_LANDED(SLOT, CNT) lgkm_landed<CNT>(ft[SLOT][0], ft[SLOT][1], ft[SLOT][2], ft[SLOT][3]);
        LB_TR(0, 0)
        LB_TR(1, 1) LB_LANDED(0, 4) LB_DX(0, 0)
        LB_TR(0, 2) LB_LANDED(1, 4) LB_DX(1, 1)
        LB_TR(1, 3) LB_LANDED(0, 4) LB_DX(0, 2)
        LB_TR(0, 4) LB_LANDED(1, 4) LB_DX(1, 3)
        LB_TR(1, 5) LB_LANDED(0, 4) LB_DX(0, 4)
        LB_TR(0, 6) LB_LANDED(1, 4) LB_DX(1, 5)
        LB_TR(1, 7) LB_LANDED(0, 4) LB_DX(0, 6)
        LB_LANDED(1, 0) LB_DX(1, 7)
#undef LB_LANDED
#undef LB_TR
#undef LB_DX
      }

      __builtin_amdgcn_sched_barrier(0);
      // ---- epilogue ---------------------------------------------------------------------------------------------------
      if constexpr (EPI == LB_LN) {
        float g[16];
        const bool odd_row = (lane & 16) != 0;
        const int r_mine = lane & 15;
        float mine = 0.f;                              // the row sum this lane hands in: row r_mine of its 16-lane row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool live = full || r0 + lb_rho(r, h) < p.N;
          const float y = fmaf(xv[r], gam, bet);
          const float gy = (live && y > 0.f) ? accx[r] * ux : 0.f;
          dgam = fmaf(gy, xv[r], dgam);
          dbet += gy;
          g[r] = gy * gam;
          // sums over the 32 lanes of the half: the two 16-lane rows are folded by ONE exchange -- even rows keep g and
          // send g xhat, odd rows the other way round, so that even rows end up with sum g and odd rows with
          // sum g xhat -- then a rotation all-reduce inside the rows (no exec masking in this loop: the 16 chains are
          // independent and interleave)
          const float gx = g[r] * xv[r];
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(
              __builtin_bit_cast(int, odd_row ? g[r] : gx), 0x401f));          // lane ^ 16
          float u = (odd_row ? gx : g[r]) + recv;
          u += lb_row_ror<8>(u); u += lb_row_ror<4>(u); u += lb_row_ror<2>(u); u += lb_row_ror<1>(u);
          mine = (r_mine == r) ? u : mine;
        }
        // value index = lane: row16 = lane >> 4 is (sum g, h = 0), (sum g xhat, h = 0), (sum g, h = 1), (sum g xhat, h = 1)
        ex[wave * 64 + lane] = mine;
        // (the rstd copy of this stage was requested at the top of the iteration, before the loads of the stages ahead)
        lb_wait_vm<8 * UPT + DMA>();                   // (only the loads of the stages ahead may still be out)
        const uint32_t ars = lds0 + kRsB + (uint32_t)(((i % 3) * kLbWaves + wave) * 256 + 16 * h);
        i32x4 rs0_ = lds_read_b128<0>(ars), rs1_ = lds_read_b128<32>(ars), rs2_ = lds_read_b128<64>(ars),
              rs3_ = lds_read_b128<96>(ars);          // (tied to the wait below as they are: a copy could be read early)
        lgkm_landed<0>(rs0_, rs1_, rs2_, rs3_);      // (and the partial sums above are in LDS)
        float rs[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rs[q] = __builtin_bit_cast(f32x4, rs0_)[q]; rs[4 + q] = __builtin_bit_cast(f32x4, rs1_)[q];
          rs[8 + q] = __builtin_bit_cast(f32x4, rs2_)[q]; rs[12 + q] = __builtin_bit_cast(f32x4, rs3_)[q];
        }
        // (bare barrier: __syncthreads() would also drain the loads and LDS-DMAs of the stages ahead)
        __builtin_amdgcn_s_barrier();
        float tot;
        {
          const uint32_t ae = lds0 + kExB + (uint32_t)lane * 4u;
          float e[8] = {lds_read_f32<0>(ae), lds_read_f32<256>(ae), lds_read_f32<512>(ae), lds_read_f32<768>(ae),
                        lds_read_f32<1024>(ae), lds_read_f32<1280>(ae), lds_read_f32<1536>(ae), lds_read_f32<1792>(ae)};
          lgkm_landed<0>(e);
          tot = ((((((e[0] + e[1]) + e[2]) + e[3]) + e[4]) + e[5]) + e[6]) + e[7];     // the 8 column strips, in order
        }
        constexpr float inv_k = 1.0f / K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s1 = __shfl(tot, 32 * h + r) * inv_k;
          const float s2 = __shfl(tot, 32 * h + 16 + r) * inv_k;
          g[r] = rs[r] * (g[r] - s1 - xv[r] * s2);
          if (!full && r0 + lb_rho(r, h) >= p.N) g[r] = 0.f;              // (rows past the end: rstd is not theirs)
          om = fmaxf(om, fabsf(g[r]));
        }
        // (uniform base + 32-bit lane offset: one address register for the 16 stores)
        char* out = reinterpret_cast<char*>(p.dx + (size_t)r0 * K);
        const uint32_t o0 = (uint32_t)(4 * h * K + col) * 4u;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            *reinterpret_cast<float*>(out + (o0 + (uint32_t)(((r & 3) + 8 * (r >> 2)) * K * 4))) = g[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (r0 + lb_rho(r, h) < p.N)
              *reinterpret_cast<float*>(out + (o0 + (uint32_t)(((r & 3) + 8 * (r >> 2)) * K * 4))) = g[r];
        }
      } else {
        // the two waves of a column strip hold the two halves of the contraction: wave mh finishes registers
        // 8 mh .. 8 mh + 7 and hands the others to its partner through LDS
        float keep[8];
        if constexpr (NM == 2) {
          float* mine = ex + wave * 512 + lane;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            keep[q] = mh ? accx[8 + q] : accx[q];
            mine[q * 64] = mh ? accx[q] : accx[8 + q];
          }
          LB_WAIT_LGKM(0);
          __builtin_amdgcn_s_barrier();             // (bare: see the LayerNorm epilogue)
          const uint32_t ae = lds0 + kExB + (uint32_t)((wave ^ NC) * 512 + lane) * 4u;
          float e[8] = {lds_read_f32<0>(ae), lds_read_f32<256>(ae), lds_read_f32<512>(ae), lds_read_f32<768>(ae),
                        lds_read_f32<1024>(ae), lds_read_f32<1280>(ae), lds_read_f32<1536>(ae), lds_read_f32<1792>(ae)};
          lgkm_landed<0>(e);
#pragma unroll
          for (int q = 0; q < 8; ++q) keep[q] += e[q];
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) keep[q] = 0.f;          // (not instantiated: every plain shape splits the m range)
        }
        if constexpr (EPI == LB_SHIFT) vm_landed<8 * UPT + DMA>(lse8);      // (only this iteration's loads ahead may be out)
        else asm volatile("" : "+v"(keep[0]), "+v"(keep[1]), "+v"(keep[2]), "+v"(keep[3]), "+v"(keep[4]), "+v"(keep[5]),
                          "+v"(keep[6]), "+v"(keep[7])::"memory");        // (the same fence for the scheduler)
        float v[8], gt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          v[q] = keep[q] * ux;
          om = fmaxf(om, fabsf(v[q]));
          if constexpr (EPI == LB_SHIFT) {
            // (a node without incoming edges is never gathered, and the forward writes lse = 0 for it)
            gt[q] = v[q] * fast_exp2(-lse8[q]);
            if (shift) worst = fmaxf(worst, fabsf(lse8[q]));
          }
        }
        char* out = reinterpret_cast<char*>(p.dx + (size_t)r0 * K);
        char* out_t = reinterpret_cast<char*>(p.gt + (size_t)r0 * K);
        const uint32_t o0 = (uint32_t)((16 * mh + 4 * h) * K + col) * 4u;
        if (full) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            *reinterpret_cast<float*>(out + (o0 + (uint32_t)(((q & 3) + 8 * (q >> 2)) * K * 4))) = v[q];
          if (EPI == LB_SHIFT && shift) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              *reinterpret_cast<float*>(out_t + (o0 + (uint32_t)(((q & 3) + 8 * (q >> 2)) * K * 4))) = gt[q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (r0 + lb_rho(8 * mh + q, h) < p.N) {
              const uint32_t o = o0 + (uint32_t)(((q & 3) + 8 * (q >> 2)) * K * 4);
              *reinterpret_cast<float*>(out + o) = v[q];
              if (EPI == LB_SHIFT && shift) *reinterpret_cast<float*>(out_t + o) = gt[q];
            }
          }
        }
      }

      // ---- the next stage of go into the other plane buffer; x of the next stage has to have landed ------------------
      __builtin_amdgcn_sched_barrier(0);
      // the go registers of stage i + 1: behind them only this iteration's LDS-DMAs and stores (a ragged stage may
      // have skipped stores: it waits for everything)
      if (full && shift) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) vm_landed<DMA + E_S>(gr[q]);
      } else if (full) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) vm_landed<DMA + (EPI == LB_SHIFT ? 8 : E_S)>(gr[q]);      // (no gt stores)
      } else {
#pragma unroll
        for (int q = 0; q < UPT; ++q) vm_landed<0>(gr[q]);
      }
      commit_go(buf ^ 1, row_of(i + 1), i + 1 < n);
      // x of stage i + 1 (requested one iteration ago): everything older than this iteration's own operations
      if (full && shift) lb_wait_vm<kVmPerIter>();
      else if (full) lb_wait_vm<kVmPerIter - (EPI == LB_SHIFT ? 8 : 0)>();
      else lb_wait_vm<0>();
      LB_WAIT_LGKM(0);                              // (the plane writes above)
      __builtin_amdgcn_s_barrier();
    }
  }
  lb_wait_vm<0>();

  // ---- partial results of this workgroup ------------------------------------------------------------------------------
  float* slab = p.ws + (size_t)blockIdx.x * p.slab_cols;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      slab[(size_t)(mh * MW + 32 * t + lb_rho(r, h)) * K + col] = accw[t][r] * uw;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < UPT; ++q) {
    const int u = tid + q * kLbThreads;
    ex[(u / M) * M + ucol(q)] = bsum[q];
  }
  if constexpr (EPI == LB_LN) {
    dgam += __shfl_xor(dgam, 32);
    dbet += __shfl_xor(dbet, 32);
    if (h == 0) { slab[M * K + M + col] = dgam; slab[M * K + M + K + col] = dbet; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { om = fmaxf(om, __shfl_xor(om, off)); worst = fmaxf(worst, __shfl_xor(worst, off)); }
  if (lane == 0) ex[4 * M + wave] = om;
  __syncthreads();
  for (int m = tid; m < M; m += kLbThreads) slab[M * K + m] = (ex[m] + ex[M + m]) + (ex[2 * M + m] + ex[3 * M + m]);
  if (tid == 0 && p.out_parts) {
    float m = ex[4 * M];
#pragma unroll
    for (int i = 1; i < kLbWaves; ++i) m = fmaxf(m, ex[4 * M + i]);
    p.out_parts[blockIdx.x] = p.parts_max_with_old ? fmaxf(m, p.out_parts[blockIdx.x]) : m;
  }
  if constexpr (EPI == LB_SHIFT) {
    // (a NaN lse fails the comparison, like in softmax_shift_kernel: the NaN then travels in gt itself)
    if (shift && lane == 0 && worst > kMaxLse) *p.spread = 1;   // plain store: every writer stores the same value
  }
}

// row maxima [n] -> kLbParts partial maxima (slices), one or two arrays per launch
__global__ __launch_bounds__(256) void lb_rowmax_parts_kernel(const float* __restrict__ a, float* __restrict__ pa,
                                                             const float* __restrict__ b, float* __restrict__ pb, int n,
                                                             int* __restrict__ zero4, float* __restrict__ zero_parts) {
  __shared__ float red[2][4];
  // (what the main kernel expects cleared -- the shift flag, the unused tail of the max table -- is cleared here
  // instead of by memset launches of their own)
  if (blockIdx.x == 0 && threadIdx.x < 4 && zero4) zero4[threadIdx.x] = 0;
  if (threadIdx.x == 0 && zero_parts) zero_parts[blockIdx.x] = 0.f;
  const int per = (n + kLbParts - 1) / kLbParts;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  float ma = 0.f, mb = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) { if (a) ma = fmaxf(ma, a[i]); if (b) mb = fmaxf(mb, b[i]); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, off)); mb = fmaxf(mb, __shfl_xor(mb, off)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ma; red[1][threadIdx.x >> 6] = mb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (a) pa[blockIdx.x] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    if (b) pb[blockIdx.x] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}

template <int M, int K>
constexpr int lb_lds_bytes() {
  return 2 * (2 * 4 * (M + 4) * 16) + 3 * (kLbStage * K * 4) + (4 * M + 64) * 4 +
         (M == 256 ? kLbWaves * 512 * 4 : kLbWaves * 64 * 4) +
         3 * kLbWaves * 256;
}

}  // namespace mlgnn

using namespace mlgnn;

static bool lb_shape_ok(int64_t N, int64_t M, int64_t K, int epi) {
  if (N <= 0 || N > INT32_MAX - kLbStage) return false;
  // (32-bit byte offsets inside the kernel: the entry point walks row slabs below 4 GiB, dense_slab_rows)
  if (epi == LB_LN) return M == 128 && K == 256;
  return M == 256 && K == 128;
}

extern "C" int mlgnn_linear_bwd_supported(int64_t N, int64_t M, int64_t K, int epilogue) {
  if (epilogue != LB_LN && epilogue != LB_PLAIN && epilogue != LB_SHIFT) return 0;
  return lb_shape_ok(N, M, K, epilogue) ? 1 : 0;
}

// workspace (floats): one partial slab per workgroup, then 3 x kLbParts partial maxima
extern "C" int64_t mlgnn_linear_bwd_workspace_floats(int64_t N, int64_t M, int64_t K, int epilogue) {
  if (!mlgnn_linear_bwd_supported(N, M, K, epilogue)) return MLGNN_E_SHAPE;
  const int64_t cols = M * K + M + (epilogue == LB_LN ? 2 * K : 0);
  return (int64_t)kLbParts * cols + 3 * kLbParts;
}

extern "C" int mlgnn_linear_bwd(const float* go, const float* w, const float* x, const float* go_max, int go_max_is_parts,
                                const float* x_row_max, int epilogue, const float* rstd, const float* gamma,
                                const float* beta, const float* lse, float* dx, float* grad_shifted,
                                int32_t* shift_flag, float* grad_w_b, float* dx_max_parts, float* workspace,
                                int64_t workspace_floats, int64_t N, int64_t M, int64_t K, void* stream) {
  if (epilogue != LB_LN && epilogue != LB_PLAIN && epilogue != LB_SHIFT) return MLGNN_E_MODE;
  if (!lb_shape_ok(N, M, K, epilogue)) return MLGNN_E_SHAPE;
  if (!go || !w || !x || !go_max || !x_row_max || !dx || !grad_w_b || !workspace) return MLGNN_E_NULL;
  if (epilogue == LB_LN && (!rstd || !gamma || !beta)) return MLGNN_E_NULL;
  if (epilogue == LB_SHIFT && (!lse || !grad_shifted || !shift_flag)) return MLGNN_E_NULL;
  if (workspace_floats < mlgnn_linear_bwd_workspace_floats(N, M, K, epilogue)) return MLGNN_E_WORKSPACE;
  if (((reinterpret_cast<uintptr_t>(go) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(x) |
        reinterpret_cast<uintptr_t>(rstd)) & 15) != 0)
    return MLGNN_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int cols = (int)(M * K + M + (epilogue == LB_LN ? 2 * K : 0));
  float* parts = workspace + (int64_t)kLbParts * cols;           // [go | x | (unused)]
  const int64_t stages = (N + kLbStage - 1) / kLbStage;
  const int grid = (int)(stages < kLbParts ? stages : kLbParts);
  hipLaunchKernelGGL(lb_rowmax_parts_kernel, dim3(kLbParts), dim3(256), 0, s, go_max_is_parts ? nullptr : go_max, parts,
                     x_row_max, parts + kLbParts, (int)N, epilogue == LB_SHIFT ? shift_flag : nullptr,
                     (dx_max_parts && grid < kLbParts) ? dx_max_parts : nullptr);
  int err;
  LbArgs a;
  a.w = w; a.go_parts = go_max_is_parts ? go_max : parts; a.x_parts = parts + kLbParts;
  a.ws = workspace; a.out_parts = dx_max_parts;
  a.gamma = gamma; a.beta = beta;
  a.spread = shift_flag;
  a.slab_cols = cols;
#define MLGNN_LB_LAUNCH(M_, K_, EPI_)                                                                        \
  {                                                                                                          \
    constexpr int lds_bytes = lb_lds_bytes<M_, K_>();                                                        \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bwd_kernel<M_, K_, EPI_>),               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                        \
    hipLaunchKernelGGL((linear_bwd_kernel<M_, K_, EPI_>), dim3(grid), dim3(kLbThreads), lds_bytes, s, a);    \
  }
  if (epilogue == LB_PLAIN) { lse = nullptr; grad_shifted = nullptr; a.spread = nullptr; }
  // row slabs below 4 GiB (one at BASELINE configs[1]; two at 5.12 M rows): operand bases advance, the operand scales stay
  // those of the whole input, a slab's partial sums are added to its predecessors' in slab order
  const int64_t slab_rows = dense_slab_rows(M > K ? M : K);
  for (int64_t r0 = 0; r0 < N; r0 += slab_rows) {
    const int64_t n = N - r0 < slab_rows ? N - r0 : slab_rows;
    const int64_t st = (n + kLbStage - 1) / kLbStage;
    const int grid = (int)(st < kLbParts ? st : kLbParts);
    a.go = go + r0 * M; a.x = x + r0 * K; a.dx = dx + r0 * K;
    a.rstd = rstd ? rstd + r0 : nullptr;
    a.lse = lse ? lse + r0 * K : nullptr; a.gt = grad_shifted ? grad_shifted + r0 * K : nullptr;
    a.N = (int)n; a.parts_max_with_old = r0 > 0;
    if (epilogue == LB_LN) MLGNN_LB_LAUNCH(128, 256, LB_LN)
    else MLGNN_LB_LAUNCH(256, 128, LB_SHIFT)            // (plain = the same instantiation without lse / grad_shifted)
    err = (int)hipGetLastError();
    if (err) return err;
    launch_reduce_partials(workspace, grad_w_b, grid, cols, s, r0 > 0);
  }
#undef MLGNN_LB_LAUNCH
  return (int)hipGetLastError();
}
