// Large bf16 "NT" GEMM of the DiffPool contraction at BASELINE configs[4] size (csrc/gemm_nt.hip).
#pragma once
#include "common.h"

namespace mlgnn {

constexpr int kGemmTile = 128;       // output tile edge (rows and columns)
constexpr int kGemmBK = 64;          // contraction depth of one LDS stage
constexpr int kGemmMaxSeg = 4;

// One term of  C = sum_s A_s B_s^T : A_s [M, K_s] and B_s [N, K_s], both row-major with the contraction
// index contiguous (leading dimensions in elements), K_s a multiple of 64.
struct GemmSeg {
  const uint16_t* a;
  const uint16_t* b;
  int64_t lda, ldb;
  int K;
  int64_t sa, sb;            // grouped launch: element offset of operand A / B from one problem of the batch to the next
};

struct GemmDesc {
  GemmSeg seg[kGemmMaxSeg];
  int nseg;
  int M, N;                  // multiples of 128
  int splits;                // split-K factor (> 1 needs `slab`)
  float* slab;               // non-NULL: fp32 partial results [splits][M][N] and nothing else
  // outputs of the un-split product (any may be NULL)
  void* c; int64_t ldc; int c_f32;                    // C [M,N], bf16 or fp32
  uint16_t* ct; int64_t ldct;                         // C^T [N,M], bf16
  const void* aux; int64_t ldaux; int aux_f32; float alpha;   // C = acc + alpha * aux[M,N]
  const float* alpha_dev;                                     // non-NULL: alpha is read from the device instead
  const uint16_t* dot; int64_t lddot; float* dot_partial;     // dot_partial[workgroup] = sum_ij dot[i][j] * acc[i][j]
  // grouped launch (B > 1 pooled graphs of one shape, models/diff_pooling.py:59-65 on a batch): `batch` problems run as
  // grid.y of ONE launch; every pointer above advances by its stride (in elements of its own type; 0 = shared)
  int batch;
  int64_t s_slab, s_c, s_ct, s_aux, s_dot, s_part;
};

// number of workgroups (= dot partials) a descriptor launches
int gemm_nt_workgroups(const GemmDesc& d);
int gemm_nt_launch(const GemmDesc& d, hipStream_t s);

}  // namespace mlgnn
