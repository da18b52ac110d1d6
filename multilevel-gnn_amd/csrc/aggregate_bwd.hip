// Backward of the fused message + aggregation kernels (see aggregate_fwd.hip for the lane layout):
// one wavefront per SOURCE node on the transposed CSR, atomic-free; plus the fixed-order partial
// reduction shared by every kernel that sums per-workgroup partials.
#include "aggregate_common.h"
#include "aggregate_short.h"

namespace mlgnn {

// ------------------------------------------------------------------------------------------------
// backward: one wave per SOURCE node j, walking its outgoing edges (j -> i)
// ------------------------------------------------------------------------------------------------
struct BwdArgs {                      // go / x / out / efull / gx / ge are T; aux, argmax, ws fp32 / int32
  const void* go; const void* x; const void* out; const float* aux; const int* argmax;
  const int* rowptr_t; const int* col_t; const int* pos_t; const int* rowptr;
  const float* ew_t; const float* eu; const float* ev; const void* efull; const int* eid_t;
  const int* geid_t;                                        // row of grad_efull per edge (NULL: eid_t)
  void* gx; void* ge; float* ws;
  const void* gt; const int* spread;                        // softmax one-row path (see softmax_shift_kernel)
  const uint8_t* slot8;                                     // max: winner's slot inside its row, 1 byte (max_slot_kernel)
  const float* t_dev; const float* p_dev;
  int N; int d; int lpr_log2; int mean; int learn_t; int add_root; int ge_accumulate;
  float t; float p; float eps;
  int cap; const int* vrows; const int* vcount;             // long source rows, as in FwdArgs (csrc/hub.hip)
  // LNB: x was y = relu?(LayerNorm(h)) -- the res+ block's pre-conv norm (deepergcn.py:236-241) -- and the row epilogue
  // takes the finished grad_y row through that LayerNorm's backward: gx receives d loss / d h (+ ln_extra, the gradient
  // arriving on h along the block's identity branch); per-workgroup [2, d] partials of d gamma / d beta go to ln_ws
  const float* ln_h; const float* ln_mean; const float* ln_rstd; const float* ln_gamma; const float* ln_beta;
  const float* ln_extra; float* ln_rowmax; float* ln_ws; int ln_relu;
};

// VEC one-byte winner slots -> ints
template <int VEC>
__device__ __forceinline__ void load_slots(int (&r)[VEC], const uint8_t* p) {
  if constexpr (VEC == 4) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (int)((w >> (8 * i)) & 0xffu);
  } else if constexpr (VEC == 8) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (int)((w.x >> (8 * i)) & 0xffu); r[4 + i] = (int)((w.y >> (8 * i)) & 0xffu); }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = (int)p[i];
  }
}

// SHIFT (softmax without a learnable temperature): the normaliser is folded into the cotangent by
// softmax_shift_kernel,  gt[i][c] = go[i][c] * 2^(-lse[i][c]),  so that  w_e * go = 2^(t m_e) * gt[i][c]  and an
// edge gathers ONE row (gt) instead of two (go, lse): half the gather traffic, and no per-edge scalar either.
// SHIFT with max: the winning edge of (i, c) is looked up as a 1-byte slot inside row i (max_slot_kernel) instead of
// the 4-byte by-destination position the forward wrote -- the second gathered row shrinks from 4 d to d bytes.
template <typename T, int VEC, int MODE, int AGGR, bool LEARN_T, bool SHIFT, bool VIRT, bool LNB = false, bool WIDE = false>
__device__ __forceinline__ void csr_aggregate_bwd_body(const BwdArgs& a, float (*red)[kWave * VEC]) {
  static_assert(!LNB || (sizeof(T) == 4 && !VIRT && !LEARN_T), "LayerNorm-backward epilogue: fp32 rows of the main launch");
  constexpr int RK = rank_of<MODE>();
  constexpr int ES = edge_scalars<MODE>();
  constexpr int ESA = ES > 0 ? ES : 1;
  const T* GO = static_cast<const T*>(a.go);
  const T* GT = static_cast<const T*>(a.gt);
  const T* X = static_cast<const T*>(a.x);
  const T* OUTS = static_cast<const T*>(a.out);
  const T* EF = static_cast<const T*>(a.efull);
  T* GX = static_cast<T*>(a.gx);
  T* GE = static_cast<T*>(a.ge);
  constexpr uint32_t kWide = 4u / (uint32_t)sizeof(T);      // fp32 side arrays (lse, argmax): byte offset scale
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int lpr = 1 << a.lpr_log2;
  const int groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2;
  const int cl = lane & (lpr - 1);
  const RowWalk walk = make_row_walk(VIRT ? *a.vcount : a.N);
  const Scalars sc = read_scalars(a.t_dev, a.p_dev, a.t, a.p);
  const float t_eps = sc.t_log2e * a.eps;
  const uint32_t row_bytes = (uint32_t)a.d * (uint32_t)sizeof(T);
  // max with a TABLE edge term (ge_accumulate == 2): only the winning edge of (i, c) has a gradient -- 1 / degree of the
  // [E, d] per-edge gradient is non-zero -- so it is added straight to the table's fixed-point accumulator (integer
  // atomics: order-independent) instead of being written per edge and reduced later (csrc/embedding.hip)
  constexpr bool kCanFix = MODE == M_GEN_FULL && AGGR == A_MAX && sizeof(T) == 4;
  const bool fix = kCanFix && a.ge_accumulate == 2;
  // ge_accumulate == 3: the table's gradient is taken from the destination side (mlgnn_max_table_grad, csrc/embedding.hip);
  // nothing per edge leaves this kernel
  const bool skip_ge = kCanFix && a.ge_accumulate == 3;
  const float fix_scale = fix ? fix_scale_of(static_cast<const uint32_t*>(a.ge)) : 0.f;
  unsigned long long* fix_tab = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(a.ge) + kFixHeaderBytes);

  for (int cbase = 0; cbase < a.d; cbase += lpr * VEC) {
    const bool cact = cbase + cl * VEC < a.d;          // inactive lanes shadow the last chunk (see forward)
    const int c0 = min(cbase + cl * VEC, a.d - VEC);
    const uint32_t c_bytes = (uint32_t)c0 * (uint32_t)sizeof(T);
    float eu[VEC][ESA], ev[VEC], gu[VEC][ESA], gv[VEC];   // eu[i][k] = U[k][c0 + i]
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      ev[i] = 0.f; gv[i] = 0.f;
#pragma unroll
      for (int k = 0; k < ESA; ++k) { eu[i][k] = 0.f; gu[i][k] = 0.f; }
    }
    if constexpr (RK > 0) {
      load_vec<VEC>(ev, a.ev + c0);
#pragma unroll
      for (int k = 0; k < RK; ++k) {
        float row[VEC];
        load_vec<VEC>(row, a.eu + (size_t)k * a.d + c0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) eu[i][k] = row[i];
      }
    }

    // LNB (one channel chunk: d == lpr * VEC, checked on the host).  The epilogue's arithmetic is the same for every
    // lane group, and after the merge below every group holds the finished row -- so a finished row is only PARKED with
    // one group (group `slot`), and the epilogue runs once per `groups` rows with each group working on its own row:
    // 1/groups of the instructions and shuffle trips per row (this kernel is bound by instruction issue; an epilogue
    // per row cost as much as the LayerNorm pass it replaces).  Only the parked row stays in registers between rows
    // (this kernel runs four waves per SIMD at <= 128 registers): the LayerNorm operands are requested at the flush,
    // the d gamma / d beta sums of the rows a lane owned live in LDS (`red`, a slot per lane and channel).
    const float inv_d = 1.0f / (float)a.d;
    int slot = 0, keep_r = -1;
    float keep[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) keep[i] = 0.f;
    float* lds_dg = &red[wave][0] + lane * VEC;                   // [kWave * VEC] per wave: this lane's d gamma ...
    float* lds_db = lds_dg + kWavesPerBlock * kWave * VEC;        // ... and d beta sums (second half of `red`)
    if constexpr (LNB) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { lds_dg[i] = 0.f; lds_db[i] = 0.f; }
    }
    // the LayerNorm operands of the rows a flush will work on, requested when the LAST row of a group of `groups`
    // starts (every lane group for the row parked with it, the last group for the row about to be walked), so that
    // they arrive behind that row's edges
    float lh[VEC], le[VEC], lroot[VEC], lmu = 0.f, lrs = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { lh[i] = 0.f; le[i] = 0.f; lroot[i] = 0.f; }
    // (read once, next to a gather that lives on its L2 hit rate: non-temporal)
    auto stream4 = [](float (&dst)[VEC], const float* src) {
      if constexpr (VEC == 4) {
        using f4 = __attribute__((ext_vector_type(4))) float;
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src));
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      } else {
        load_vec<VEC>(dst, src);
      }
    };
    auto ln_request = [&](int kr) {
      stream4(lh, a.ln_h + (size_t)kr * a.d + c0);
      lmu = a.ln_mean[kr]; lrs = a.ln_rstd[kr];
      if (a.ln_extra) stream4(le, a.ln_extra + (size_t)kr * a.d + c0);
      if (a.add_root) stream4(lroot, reinterpret_cast<const float*>(GO) + (size_t)kr * a.d + c0);
    };
    auto ln_flush = [&](bool requested) {
      const bool wr = keep_r >= 0;
      const int kr = max(keep_r, 0);
      if (!requested) ln_request(kr);
      float lng[VEC], lnb[VEC], ldg[VEC], ldb[VEC];
      load_vec<VEC>(lng, a.ln_gamma + c0);
      load_vec<VEC>(lnb, a.ln_beta + c0);
      load_vec<VEC>(ldg, lds_dg);
      load_vec<VEC>(ldb, lds_db);
      float xh[VEC], gg[VEC], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        xh[i] = (lh[i] - lmu) * lrs;
        const float y = fmaf(xh[i], lng[i], lnb[i]);
        const float gy = (wr && !(a.ln_relu && !(y > 0.f))) ? keep[i] + lroot[i] : 0.f;
        ldg[i] = fmaf(gy, xh[i], ldg[i]);
        ldb[i] += gy;
        gg[i] = gy * lng[i];
        s1 += gg[i];
        s2 = fmaf(gg[i], xh[i], s2);
      }
      store_vec<VEC>(lds_dg, ldg);
      store_vec<VEC>(lds_db, ldb);
      // d = 128: a lane group is a 32-lane half -- rotations inside the rows of 16 lanes as DPP modifiers, one
      // crossbar trip for the last step (common.h); the shuffle ladder (five trips per sum) otherwise
      if (a.lpr_log2 == 5) { s1 = half_sum(s1); s2 = half_sum(s2); }
      else for (int off = 1; off < lpr; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
      s1 *= inv_d; s2 *= inv_d;
      float o[VEC], om = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        o[i] = wr ? fmaf(lrs, gg[i] - s1 - xh[i] * s2, le[i]) : 0.f;
        om = fmaxf(om, fabsf(o[i]));
      }
      if (a.ln_rowmax) {
        if (a.lpr_log2 == 5) om = half_max(om);
        else for (int off = 1; off < lpr; off <<= 1) om = fmaxf(om, __shfl_xor(om, off));
        if (wr && cl == 0) a.ln_rowmax[kr] = om;
      }
      if constexpr (VEC == 4) {
        if (wr) {
          using f4 = __attribute__((ext_vector_type(4))) float;
          const f4 v = {o[0], o[1], o[2], o[3]};
          __builtin_nontemporal_store(v, reinterpret_cast<f4*>(reinterpret_cast<float*>(GX) + (size_t)kr * a.d + c0));
        }
      }
      keep_r = -1;
      slot = 0;
    };

    for (int r = walk.first; r < walk.r_end; r += walk.stride) {
      // xr: the source node whose features this row of edges belongs to (VIRT: a chunk of a long row, csrc/hub.hip)
      const int xr = VIRT ? a.vrows[3 * r] : r;
      const int beg = VIRT ? a.vrows[3 * r + 1] : a.rowptr_t[r];
      const int end = VIRT ? a.vrows[3 * r + 2] : min(a.rowptr_t[r + 1], beg + a.cap);
      float xj[VEC], gx[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) { xj[i] = 0.f; gx[i] = 0.f; }
      // max: the forward names a winner only where its z > 0 (argmax = -1 otherwise): z is not needed again
      if (is_gen<MODE>() && AGGR != A_MAX && end > beg) load_t<T, VEC>(xj, X + (size_t)xr * a.d + c0);
      float xjv[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) xjv[i] = xj[i] + ev[i];
      if constexpr (LNB) {
        if (slot == groups - 1) ln_request(sub == slot ? r : max(keep_r, 0));
      }
      for (int base = beg; base < end; base += kWave) {
        const int cnt = min(kWave, end - base);
        uint32_t my_off = 0;
        int my_pos = 0, my_eid = 0, my_gid = 0;
        float my_ew[ESA], my_inv = 1.f;
#pragma unroll
        for (int k = 0; k < ESA; ++k) my_ew[k] = 0.f;
        if (lane < cnt) {
          const int dst = a.col_t[base + lane];
          my_off = row_key<WIDE>(dst, row_bytes);
          if (AGGR == A_MAX) my_pos = a.pos_t[base + lane] - (SHIFT ? a.rowptr[dst] : 0);
          if constexpr (ES > 0) {
            load_edge_scalars<ES>(my_ew, a.ew_t, (size_t)(base + lane));
          }
          if (MODE == M_GEN_FULL && !skip_ge) {
            my_eid = a.eid_t[base + lane];
            my_gid = a.geid_t ? a.geid_t[base + lane] : my_eid;
          }
          if (AGGR == A_SUM && a.mean)
            my_inv = __builtin_amdgcn_rcpf((float)max(a.rowptr[dst + 1] - a.rowptr[dst], 1));
        }

        auto batch = [&](auto full_c, const int k) {
          constexpr bool FULL = decltype(full_c)::value;
          float ga[kUnroll][VEC], gb[kUnroll][VEC], gc[kUnroll][VEC], ef[kUnroll][VEC];
          int ai[kUnroll][VEC];
          float wa[kUnroll][ESA], inv[kUnroll];
          int pos[kUnroll], e0[kUnroll], g0[kUnroll];
          bool valid[kUnroll];
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const int idx = k + u * groups + sub;
            valid[u] = FULL || (idx < cnt);
            const int src = idx & (kWave - 1);
            const auto off = row_offset<WIDE>((uint32_t)__shfl((int)my_off, src), row_bytes, c_bytes);
#pragma unroll
            for (int q = 0; q < ESA; ++q) wa[u][q] = (ES > 0) ? __shfl(my_ew[q], src) : 0.f;
            inv[u] = (AGGR == A_SUM) ? __shfl(my_inv, src) : 1.f;
            pos[u] = (AGGR == A_MAX) ? __shfl(my_pos, src) : 0;
            e0[u] = (MODE == M_GEN_FULL) ? __shfl(my_eid, src) : 0;
            g0[u] = (MODE == M_GEN_FULL) ? __shfl(my_gid, src) : 0;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { ga[u][i] = 0.f; gb[u][i] = 0.f; gc[u][i] = 0.f; ef[u][i] = 0.f; ai[u][i] = -2; }
            if (FULL || valid[u]) {
              load_row<T, VEC>(ga[u], (SHIFT && AGGR == A_SOFTMAX) ? GT : GO, off);
              if (AGGR == A_SOFTMAX && !SHIFT) load_row<float, VEC>(gb[u], a.aux, off * kWide);
              if (AGGR == A_SOFTMAX && LEARN_T) load_row<T, VEC>(gc[u], OUTS, off);
              if (AGGR == A_MAX && !SHIFT) load_row<VEC>(ai[u], a.argmax, off * kWide);
              if (AGGR == A_MAX && SHIFT) load_slots<VEC>(ai[u], a.slot8 + off / (uint32_t)sizeof(T));
              if (MODE == M_GEN_FULL && AGGR != A_MAX) load_t<T, VEC>(ef[u], EF + (size_t)e0[u] * a.d + c0);
            }
          }
          // lane-group chunks of a partial batch past the row's last edge contribute nothing: skipped as a whole
          const int live = FULL ? kUnroll : (cnt - k + groups - 1) >> (6 - a.lpr_log2);
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            if (!FULL && u >= live) continue;
            float dz[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              float coef, z = 0.f, m = 0.f;
              if constexpr (AGGR == A_MAX) {        // (the winner's z > 0 by construction of argmax)
              } else if constexpr (RK > 0) {        // x_j + v is constant along the row: the fma chain starts from it
                z = xjv[i];
#pragma unroll
                for (int q = 0; q < RK; ++q) z = fmaf(wa[u][q], eu[i][q], z);
                m = fmaxf(z, 0.f) + a.eps;
              } else if constexpr (is_gen<MODE>()) {
                z = pre_act<MODE>(xj[i], wa[u], eu[i], ev[i], ef[u][i]);
                m = fmaxf(z, 0.f) + a.eps;
              }
              if constexpr (AGGR == A_SUM) {
                coef = ga[u][i] * inv[u];
              } else if constexpr (AGGR == A_MAX) {
                coef = (ai[u][i] == pos[u]) ? ga[u][i] : 0.f;
              } else if constexpr (AGGR == A_SOFTMAX) {
                // SHIFT: 2^(t (relu(z) + eps)) with the eps term folded into the fma's addend
                const float w = SHIFT ? fast_exp2(fmaf(sc.t_log2e, fmaxf(z, 0.f), t_eps))
                                      : fast_exp2(fmaf(sc.t_log2e, m, -gb[u][i]));
                coef = ga[u][i] * w;
                if (LEARN_T) coef *= fmaf(sc.t, m - gc[u][i], 1.0f);
              } else {  // POWER: ga carries q (see mlgnn.h)
                const float mc = fminf(fmaxf(m, kPowLo), kPowHi);
                const bool inr = (m >= kPowLo) && (m <= kPowHi);
                coef = inr ? ga[u][i] * fast_exp2((sc.p - 1.0f) * fast_log2(mc)) : 0.f;
              }
              if constexpr (MODE == M_WEIGHTED) coef *= wa[u][0];
              if constexpr (is_gen<MODE>() && AGGR != A_MAX) coef = (z > 0.f) ? coef : 0.f;
              dz[i] = (FULL || valid[u]) ? coef : 0.f;
              gx[i] += dz[i];
              if constexpr (RK > 0) {               // (d loss / d v = sum of all dz: taken from gx once per row, below)
#pragma unroll
                for (int q = 0; q < RK; ++q) gu[i][q] = fmaf(wa[u][q], dz[i], gu[i][q]);
              }
            }
            if constexpr (kCanFix) {
              if (fix) {
                if (valid[u] && cact) {
                  unsigned long long* tp = fix_tab + (size_t)g0[u] * a.d + c0;
#pragma unroll
                  for (int i = 0; i < VEC; ++i)
                    if (dz[i] != 0.f) atomicAdd(tp + i, (unsigned long long)__float2ll_rn(dz[i] * fix_scale));
                }
                continue;
              }
              if (skip_ge) continue;
            }
            if (MODE == M_GEN_FULL && valid[u] && cact) {
              T* gep = GE + (size_t)g0[u] * a.d + c0;
              if (a.ge_accumulate) {                 // this layer's share on top of the layers that ran before it
                if constexpr (AGGR == A_MAX) {       // max: only a row's winners carry gradient -- most pieces add nothing
                  bool any = false;
#pragma unroll
                  for (int i = 0; i < VEC; ++i) any |= dz[i] != 0.f;
                  if (!any) continue;
                }
                float prev[VEC];
                load_t<T, VEC>(prev, gep);
#pragma unroll
                for (int i = 0; i < VEC; ++i) dz[i] += prev[i];
              }
              store_t<T, VEC>(gep, dz);
            }
          }
        };

        const int step = groups * kUnroll;
        int k = 0;
        for (; k + step <= cnt; k += step) batch(BC<true>{}, k);
        if (k < cnt) batch(BC<false>{}, k);
      }
      if constexpr (RK > 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv[i] += gx[i];          // this lane's share of the row, before the groups merge
      }
      for (int off = lpr; off < kWave; off <<= 1)
#pragma unroll
        for (int i = 0; i < VEC; ++i) gx[i] += __shfl_xor(gx[i], off);
      if constexpr (LNB) {
        if (sub == slot) {
          keep_r = r;
#pragma unroll
          for (int i = 0; i < VEC; ++i) keep[i] = gx[i];
        }
        if (++slot == groups) ln_flush(true);
      } else
      if (sub == 0 && cact) {
        if (a.add_root && !VIRT) {    // identity branch of h = x + m (once per real row: the main launch)
          float gr[VEC];
          load_t<T, VEC>(gr, GO + (size_t)r * a.d + c0);
#pragma unroll
          for (int i = 0; i < VEC; ++i) gx[i] += gr[i];
        }
        if constexpr (VIRT) store_vec<VEC>(reinterpret_cast<float*>(a.gx) + (size_t)r * a.d + c0, gx);   // chunk partial: fp32
        else store_t<T, VEC>(GX + (size_t)r * a.d + c0, gx);     // (non-temporal measured here: 0.75 -> 0.85 ms per launch)
      }
    }

    if constexpr (LNB) {
      if (slot > 0) ln_flush(false);               // rows still parked when the walk ends
    }
    if constexpr (LNB) {
      // d gamma / d beta: every lane holds (in LDS) the sums of the rows its group owned -> groups -> waves -> one
      // [2, d] partial per workgroup
      __syncthreads();
      for (int c = threadIdx.x; c < 2 * a.d; c += kBlock) {
        const int which = c / a.d, ch = c % a.d;
        const float* src = &red[0][0] + which * (kWavesPerBlock * kWave * VEC);
        float sum = 0.f;
        for (int w = 0; w < kWavesPerBlock; ++w)
          for (int g2 = 0; g2 < groups; ++g2) sum += src[w * (kWave * VEC) + (g2 * lpr) * VEC + ch];
        a.ln_ws[((size_t)blockIdx.x * 2 + which) * a.d + ch] = sum;
      }
    }
    if constexpr (RK > 0) {
      // per-workgroup partial of d loss/d U [RK,d] and d loss/d v [d]  ->  ws[block][RK+1][d]; summed by a second launch
      for (int off = lpr; off < kWave; off <<= 1)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          gv[i] += __shfl_xor(gv[i], off);
#pragma unroll
          for (int k = 0; k < RK; ++k) gu[i][k] += __shfl_xor(gu[i][k], off);
        }
      // red[] is indexed by the lane's nominal column; shadow lanes land past d and are skipped below
#pragma unroll
      for (int which = 0; which <= RK; ++which) {
        __syncthreads();
        if (sub == 0) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) red[wave][cl * VEC + i] = (which < RK) ? gu[i][which < RK ? which : 0] : gv[i];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < lpr * VEC; c += kBlock) {
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < kWavesPerBlock; ++w) s += red[w][c];
          if (cbase + c < a.d) a.ws[((size_t)blockIdx.x * (RK + 1) + which) * a.d + cbase + c] = s;
        }
      }
    }
  }
}

// (LNB: four waves per SIMD asked for -- the parked row puts the softmax instantiation two registers past 128)
template <typename T, int VEC, int MODE, int AGGR, bool LEARN_T, bool VIRT = false, bool LNB = false, bool WIDE = false>
__global__ __launch_bounds__(kBlock, LNB ? 4 : 1) void csr_aggregate_bwd_kernel(const BwdArgs a) {
  __shared__ float red[LNB ? 2 * kWavesPerBlock : kWavesPerBlock][kWave * VEC];     // LNB: + the d gamma / d beta slots
  if constexpr (AGGR == A_SOFTMAX && !LEARN_T) {
    // *a.spread != 0: softmax_shift_kernel met a node with |lse| > kMaxLse in some channel; the two-row path
    // stays as the fallback for such inputs (never seen in practice: lse = log2 sum_e 2^(t m_e))
    const bool shift_ok = a.gt != nullptr && *a.spread == 0;
    if (shift_ok) csr_aggregate_bwd_body<T, VEC, MODE, AGGR, LEARN_T, true, VIRT, LNB, WIDE>(a, red);
    else csr_aggregate_bwd_body<T, VEC, MODE, AGGR, LEARN_T, false, VIRT, LNB, WIDE>(a, red);
  } else if constexpr (AGGR == A_MAX) {
    // *a.spread != 0: some node has more than 254 incoming edges, its slots do not fit a byte
    const bool slots_ok = a.slot8 != nullptr && *a.spread == 0;
    if (slots_ok) csr_aggregate_bwd_body<T, VEC, MODE, AGGR, LEARN_T, true, VIRT, LNB, WIDE>(a, red);
    else csr_aggregate_bwd_body<T, VEC, MODE, AGGR, LEARN_T, false, VIRT, LNB, WIDE>(a, red);
  } else {
    csr_aggregate_bwd_body<T, VEC, MODE, AGGR, LEARN_T, false, VIRT, LNB, WIDE>(a, red);
  }
}

// max: slot8[i][c] = argmax[i][c] - rowptr[i] (the winner's position inside row i; 255 = no incoming edge), one
// byte per channel; *spread is set when a row is too long for that (in-degree > 254).  Streaming: reads N d 4,
// writes N d bytes; the backward then gathers d instead of 4 d bytes of winner information per edge.
struct SlotArgs {
  const int* argmax; const int* rowptr; uint8_t* slot8; int* spread;
  int N; int d;
};

__global__ __launch_bounds__(kBlock) void max_slot_kernel(const SlotArgs a) {
  const int64_t quads = (int64_t)a.N * a.d / 4;                 // d % 4 == 0 on this path
  bool too_long = false;
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < quads; q += (int64_t)gridDim.x * kBlock) {
    const int row = (int)(q * 4 / a.d);
    const int beg = a.rowptr[row], end = a.rowptr[row + 1];
    too_long |= (end - beg) > 254;
    const int4 v = reinterpret_cast<const int4*>(a.argmax)[q];
    const int e[4] = {v.x, v.y, v.z, v.w};
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) w |= (uint32_t)(e[i] < 0 ? 255 : min(e[i] - beg, 254)) << (8 * i);
    reinterpret_cast<uint32_t*>(a.slot8)[q] = w;
  }
  if (__any(too_long) && (threadIdx.x & (kWave - 1)) == 0) *a.spread = 1;     // plain store of the same value
}

// Per destination node i with at least one incoming edge: gt[i][c] = go[i][c] * 2^(-lse[i][c]); *spread is set
// when some |lse| exceeds kMaxLse (a plain store of the same value by whoever sees it: no atomics on the common
// path).  Nodes without incoming edges are never gathered; their rows are written as zeros.
struct ShiftArgs {
  const void* go; const float* lse; const int* rowptr; void* gt; int* spread;
  int N; int d; int lpr_log2;
};

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void softmax_shift_kernel(const ShiftArgs a) {
  constexpr int kRows = 4;                           // row groups in flight per wave
  const T* GO = static_cast<const T*>(a.go);
  T* GT = static_cast<T*>(a.gt);
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << a.lpr_log2;
  const int groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2, cl = lane & (lpr - 1);
  const int wave_global = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int n_waves = gridDim.x * kWavesPerBlock;
  const bool one_chunk = a.d <= lpr * VEC;            // the whole row sits in one register chunk (d <= 256 / 512)
  const bool cact = cl * VEC < a.d;
  const int c0 = min(cl * VEC, a.d - VEC);
  float worst = 0.f;
  for (int r0 = wave_global * groups * kRows; r0 < a.N; r0 += n_waves * groups * kRows) {
    int row[kRows];
    bool live[kRows];
    float lo[kRows], hi[kRows], l[kRows][VEC], g[kRows][VEC];
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
      row[k] = r0 + k * groups + sub;
      const int rc = min(row[k], a.N - 1);
      live[k] = row[k] < a.N && a.rowptr[rc + 1] > a.rowptr[rc];
      lo[k] = 3.0e38f; hi[k] = -3.0e38f;
      if (one_chunk) {
        load_vec<VEC>(l[k], a.lse + (size_t)rc * a.d + c0);
        load_t<T, VEC>(g[k], GO + (size_t)rc * a.d + c0);
      }
    }
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
      const int rc = min(row[k], a.N - 1);
      if (one_chunk) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { lo[k] = fminf(lo[k], l[k][i]); hi[k] = fmaxf(hi[k], l[k][i]); }
      } else {
        for (int cbase = 0; cbase < a.d; cbase += lpr * VEC) {
          float t[VEC];
          load_vec<VEC>(t, a.lse + (size_t)rc * a.d + min(cbase + cl * VEC, a.d - VEC));
#pragma unroll
          for (int i = 0; i < VEC; ++i) { lo[k] = fminf(lo[k], t[i]); hi[k] = fmaxf(hi[k], t[i]); }
        }
      }
      for (int off = 1; off < lpr; off <<= 1) {
        lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
        hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
      }
      if (live[k]) worst = fmaxf(worst, fmaxf(hi[k], -lo[k]));
      if (one_chunk) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[k][i] = live[k] ? g[k][i] * fast_exp2(-l[k][i]) : 0.f;
        if (row[k] < a.N && cact) store_t<T, VEC>(GT + (size_t)row[k] * a.d + c0, g[k]);
      } else {
        for (int cbase = 0; cbase < a.d; cbase += lpr * VEC) {
          const int cc = min(cbase + cl * VEC, a.d - VEC);
          float t[VEC], q[VEC];
          load_vec<VEC>(t, a.lse + (size_t)rc * a.d + cc);
          load_t<T, VEC>(q, GO + (size_t)rc * a.d + cc);
#pragma unroll
          for (int i = 0; i < VEC; ++i) q[i] = live[k] ? q[i] * fast_exp2(-t[i]) : 0.f;
          if (row[k] < a.N && cbase + cl * VEC < a.d) store_t<T, VEC>(GT + (size_t)row[k] * a.d + cc, q);
        }
      }
    }
  }
  for (int off = 1; off < kWave; off <<= 1) worst = fmaxf(worst, __shfl_xor(worst, off));
  if (lane == 0 && worst > kMaxLse) *a.spread = 1;       // plain store: every writer stores the same value
}

// ws[nblk][cols] -> out[cols] in a fixed summation order (bitwise reproducible).  One workgroup of
// 1024 threads owns 32 columns: 32 row slices x 32 columns, each thread sums every 32nd row with
// independent (pipelined) loads, then the 32 slices are folded through LDS in slice order.
constexpr int kRedCols = 32, kRedSlices = 32;
__global__ __launch_bounds__(kRedCols * kRedSlices) void reduce_partials_kernel(const float* __restrict__ ws,
                                                                                 float* __restrict__ out,
                                                                                 int nblk, int cols, int accumulate) {
  __shared__ float part[kRedSlices][kRedCols + 1];
  const int cl = threadIdx.x % kRedCols;
  const int slice = threadIdx.x / kRedCols;
  const int c = blockIdx.x * kRedCols + cl;
  float s = 0.f;
  if (c < cols)
    for (int b = slice; b < nblk; b += kRedSlices) s += ws[(size_t)b * cols + c];
  part[slice][cl] = s;
  __syncthreads();
  if (slice == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kRedSlices; ++k) t += part[k][cl];
    out[c] = accumulate ? out[c] + t : t;          // (accumulate: a later row slab of the same reduction, csrc/wgrad.hip)
  }
}

// Wide tables (the [M K] weight-gradient partials of csrc/linear_bwd.hip / wgrad.hip: 256 slabs x 33 000 columns):
// 16-byte loads, 1 KB of a slab row per workgroup pass (the narrow kernel reads 128-byte pieces), 4 row slices of 64
// column quads, slices folded in order.
constexpr int kRedWQuads = 64, kRedWSlices = 4;
__global__ __launch_bounds__(kRedWQuads * kRedWSlices) void reduce_partials_wide_kernel(const float4* __restrict__ ws,
                                                                                       float4* __restrict__ out,
                                                                                       int nblk, int quads, int accumulate) {
  __shared__ float4 part[kRedWSlices][kRedWQuads];
  const int ql = threadIdx.x % kRedWQuads;
  const int slice = threadIdx.x / kRedWQuads;
  const int q = blockIdx.x * kRedWQuads + ql;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < quads) {
#pragma unroll 4
    for (int b = slice; b < nblk; b += kRedWSlices) {
      const float4 v = ws[(size_t)b * quads + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[slice][ql] = s;
  __syncthreads();
  if (slice == 0 && q < quads) {
    float4 t = part[0][ql];
#pragma unroll
    for (int k = 1; k < kRedWSlices; ++k) {
      const float4 v = part[k][ql];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    if (accumulate) { const float4 o = out[q]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    out[q] = t;
  }
}

// Narrow tables with many rows (LayerNorm / edge-term parameter gradients: 512 .. 2048 partial rows of 256 .. 1024
// columns): the 32-columns-per-workgroup kernel above runs on cols / 32 = 8 .. 32 workgroups, i.e. on a tenth of the
// chip (19 us for an 8 MB table at BASELINE configs[4]).  Here a workgroup owns CQ column quads (16-byte loads) and
// 1024 / CQ row slices; CQ is picked so that the launch has >= 128 workgroups.  Fixed summation order: a thread's rows
// in order, then the slices in two levels (32 groups in order, the groups in order).
template <int CQ>
__global__ __launch_bounds__(1024) void reduce_partials_quads_kernel(const float4* __restrict__ ws, float4* __restrict__ out,
                                                                    int nblk, int quads, int accumulate) {
  constexpr int kSlices = 1024 / CQ;
  __shared__ float4 part[kSlices][CQ];
  __shared__ float4 part2[32][CQ];
  const int ql = threadIdx.x % CQ, slice = threadIdx.x / CQ;
  const int q = blockIdx.x * CQ + ql;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < quads) {
#pragma unroll 4
    for (int b = slice; b < nblk; b += kSlices) {
      const float4 v = ws[(size_t)b * quads + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[slice][ql] = s;
  __syncthreads();
  if (threadIdx.x < 32 * CQ) {
    constexpr int kPer = kSlices / 32;
    const int g = threadIdx.x / CQ;
    float4 t = part[g * kPer][ql];
#pragma unroll
    for (int k = 1; k < kPer; ++k) {
      const float4 v = part[g * kPer + k][ql];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    part2[g][ql] = t;
  }
  __syncthreads();
  if (threadIdx.x < CQ && q < quads) {
    float4 t = part2[0][ql];
#pragma unroll
    for (int k = 1; k < 32; ++k) {
      const float4 v = part2[k][ql];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    if (accumulate) { const float4 o = out[q]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    out[q] = t;
  }
}

void launch_reduce_partials(const float* ws, float* out, int nblk, int cols, hipStream_t stream, bool accumulate) {
  const int acc = accumulate ? 1 : 0;
  const bool vec_ok = cols % 4 == 0 && ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (cols >= 4096 && vec_ok) {
    const int quads = cols / 4;
    hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((quads + kRedWQuads - 1) / kRedWQuads),
                       dim3(kRedWQuads * kRedWSlices), 0, stream, reinterpret_cast<const float4*>(ws),
                       reinterpret_cast<float4*>(out), nblk, quads, acc);
    return;
  }
  static const bool quads_on = [] { const char* e = getenv("MLGNN_RED_QUADS"); return !(e && e[0] == '0'); }();   // A/B switch
  if (quads_on && vec_ok && cols >= 64 && nblk >= 256) {
    const int quads = cols / 4;
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    float4* o4 = reinterpret_cast<float4*>(out);
    if (quads >= 128 * 8)
      hipLaunchKernelGGL(reduce_partials_quads_kernel<8>, dim3((quads + 7) / 8), dim3(1024), 0, stream, w4, o4, nblk, quads, acc);
    else if (quads >= 128 * 4)
      hipLaunchKernelGGL(reduce_partials_quads_kernel<4>, dim3((quads + 3) / 4), dim3(1024), 0, stream, w4, o4, nblk, quads, acc);
    else
      hipLaunchKernelGGL(reduce_partials_quads_kernel<2>, dim3((quads + 1) / 2), dim3(1024), 0, stream, w4, o4, nblk, quads, acc);
    return;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((cols + kRedCols - 1) / kRedCols), dim3(kRedCols * kRedSlices),
                     0, stream, ws, out, nblk, cols, acc);
}

}  // namespace mlgnn

using namespace mlgnn;

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

extern "C" int64_t mlgnn_csr_aggregate_bwd_slots_offset_floats(int64_t N, int64_t d, int edge_rank) {
  if (N <= 0 || d <= 0 || d % 4 != 0) return MLGNN_E_SHAPE;
  if (edge_rank != 0 && edge_rank != 1 && edge_rank != 2 && edge_rank != 4 && edge_rank != 8) return MLGNN_E_MODE;
  const int rk = edge_rank;
  return rk > 0 ? (int64_t)(grid_for_rows(N) + kHubBlocks) * (rk + 1) * d : 0;
}

static int csr_aggregate_bwd_impl(const void* grad_out, const void* x, const void* out, const float* aux,
                                       const int32_t* argmax,
                                       const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                                       const int32_t* rowptr,
                                       const float* ew_t, const float* eu, const float* ev,
                                       const void* efull, const int32_t* eid_t, const int32_t* geid_t,
                                       void* grad_x, void* grad_efull, float* grad_uv,
                                       float* workspace, int64_t workspace_floats,
                                       int64_t N, int64_t d, int dtype, int msg, int edge_mode, int edge_rank,
                                       int aggr, int learn_t, float t, float p, const float* t_dev,
                                       const float* p_dev, float eps, int add_root, int accumulate_efull,
                                       const mlgnn_hub_t* hub, const void* grad_shifted, const int32_t* shift_flag,
                                       const mlgnn_ln_fold_t* ln, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (N < 0 || d <= 0 || N > INT32_MAX || d > INT32_MAX) return MLGNN_E_SHAPE;
  const int mode = pick_mode(msg, edge_mode, edge_rank);
  const int ag = pick_aggr(aggr);
  if (mode < 0 || ag < 0) return MLGNN_E_MODE;
  if (!is_gen_mode(mode) && ag != A_SUM) return MLGNN_E_MODE;
  if (N == 0) return 0;
  if (!grad_out || !rowptr_t || !grad_x) return MLGNN_E_NULL;   // col_t may be NULL iff E == 0
  if (is_gen_mode(mode) && !x) return MLGNN_E_NULL;
  if (aggr == MLGNN_AGGR_MEAN && !rowptr) return MLGNN_E_NULL;
  if (ag == A_MAX && !argmax) return MLGNN_E_NULL;
  if (ag == A_SOFTMAX && (!aux || (learn_t && !out))) return MLGNN_E_NULL;
  const int rk = rank_of_mode(mode);
  if ((mode == M_WEIGHTED || rk > 0) && !ew_t && col_t) return MLGNN_E_NULL;
  if (rk > 0 && (!eu || !ev || !grad_uv)) return MLGNN_E_NULL;
  if (mode == M_GEN_FULL && col_t && (!efull || !eid_t || (!grad_efull && accumulate_efull != 3))) return MLGNN_E_NULL;
  const int nblk = grid_for_rows(N);
  const bool bf16 = dtype == MLGNN_DTYPE_BF16;
  const bool split = hub && hub->cap > 0 && col_t;
  if (split) {
    if (!hub->vrows || !hub->hubs || !hub->counts || !hub->tmp) return MLGNN_E_NULL;
    if (hub->capacity < 1 || hub->tmp_bytes < mlgnn_hub_scratch_bytes(hub->capacity, d)) return MLGNN_E_WORKSPACE;
    if ((int64_t)hub->capacity * d * 4 >= (int64_t)1 << 32) return MLGNN_E_SHAPE;
  }
  // workspace = [edge-term partials: (nblk + kHubBlocks) * (rk+1) * d][softmax one-row path: flag (4 floats), gt [N*d] of T]
  const int64_t part_floats = rk > 0 ? (int64_t)(nblk + kHubBlocks) * (rk + 1) * d : 0;
  // the shifted cotangent may arrive ready-made from the producer of grad_out (mlgnn_tallgemm_nt_shift)
  const bool have_shift = ag == A_SOFTMAX && !learn_t && grad_shifted != nullptr && shift_flag != nullptr;
  const bool want_shift = ag == A_SOFTMAX && !learn_t && !have_shift;
  const bool want_slots = ag == A_MAX && d % 4 == 0;
  const int64_t shift_floats = want_shift ? 4 + (N * d * (bf16 ? 2 : 4) + 3) / 4 : (want_slots ? 4 + (N * d + 3) / 4 : 0);
  if (part_floats + shift_floats > 0 && (!workspace || workspace_floats < part_floats + shift_floats)) return MLGNN_E_WORKSPACE;

  BwdArgs a;
  a.go = grad_out; a.x = x; a.out = out; a.aux = aux;
  a.argmax = argmax; a.rowptr_t = rowptr_t; a.col_t = col_t; a.pos_t = pos_t; a.rowptr = rowptr;
  a.ew_t = ew_t; a.eu = eu; a.ev = ev; a.efull = efull; a.eid_t = eid_t; a.geid_t = geid_t;
  a.gx = grad_x; a.ge = grad_efull; a.ws = workspace;
  a.N = (int)N; a.d = (int)d; a.mean = (aggr == MLGNN_AGGR_MEAN); a.learn_t = learn_t;
  a.t = t; a.p = p; a.eps = eps; a.t_dev = t_dev; a.p_dev = p_dev; a.add_root = add_root;
  a.ge_accumulate = accumulate_efull;
  // 2: grad_efull is the fixed-point accumulator of a TABLE gradient (mlgnn_table_grad_begin), geid_t names every
  // edge's table row -- the max aggregator over fp32 rows only
  if (accumulate_efull == 3 && (ag != A_MAX || mode != M_GEN_FULL || bf16 || grad_efull)) return MLGNN_E_MODE;
  if (accumulate_efull == 2 && (ag != A_MAX || mode != M_GEN_FULL || bf16 || !grad_efull || !geid_t)) return MLGNN_E_MODE;
  a.cap = split ? hub->cap : kNoCap; a.vrows = nullptr; a.vcount = nullptr;
  if (add_root && learn_t) return MLGNN_E_MODE;       // `out` must be the bare aggregate for d/dt

  const bool al = aligned16(grad_out) && aligned16(grad_x) && (!x || aligned16(x)) &&
                  (!out || aligned16(out)) && (!aux || aligned16(aux)) && (!argmax || aligned16(argmax)) &&
                  (!efull || aligned16(efull)) && (!grad_efull || aligned16(grad_efull)) &&
                  (!eu || aligned16(eu)) && (!ev || aligned16(ev));
  {
    const uintptr_t need = rk >= 4 ? 16 : 4 * (uintptr_t)(rk > 0 ? rk : 1);     // vector loads of the edge scalar table
    if (ew_t && (reinterpret_cast<uintptr_t>(ew_t) % need) != 0) return MLGNN_E_ALIGN;
  }
  const int vec = bf16 ? ((d % 8 == 0 && al) ? 8 : 1) : ((d % 4 == 0 && al) ? 4 : 1);
  const dim3 block(kBlock);
  int launched_blocks = nblk;
  hipStream_t s = (hipStream_t)stream;
  a.lpr_log2 = lanes_per_row_log2(d, vec);
  const bool wide = needs_wide_rows(N, d) || force_wide_rows();      // 64-bit row addresses (aggregate_common.h)
  if (wide && vec == 1 && needs_wide_rows(N, d)) return MLGNN_E_SHAPE;
  a.ln_h = nullptr; a.ln_mean = a.ln_rstd = a.ln_gamma = a.ln_beta = a.ln_extra = nullptr;
  a.ln_rowmax = nullptr; a.ln_ws = nullptr; a.ln_relu = 0;
  if (ln) {
    // the epilogue holds whole fp32 rows in one lane group; long rows (whose gradient is finished by the combine
    // launch) and the d/dt path keep the separate LayerNorm backward
    if (bf16 || vec != 4 || d != ((int64_t)4 << a.lpr_log2) || split || learn_t || wide) return MLGNN_E_MODE;
    if (!ln->h || !ln->mean || !ln->rstd || !ln->gamma || !ln->beta || !ln->grad_gamma_beta || !ln->workspace) return MLGNN_E_NULL;
    if (ln->workspace_floats < (int64_t)nblk * 2 * d) return MLGNN_E_WORKSPACE;
    if (!aligned16(ln->h) || !aligned16(ln->gamma) || !aligned16(ln->beta) || (ln->grad_extra && !aligned16(ln->grad_extra)))
      return MLGNN_E_ALIGN;
    a.ln_h = ln->h; a.ln_mean = ln->mean; a.ln_rstd = ln->rstd; a.ln_gamma = ln->gamma; a.ln_beta = ln->beta;
    a.ln_extra = ln->grad_extra; a.ln_rowmax = ln->row_max; a.ln_ws = ln->workspace; a.ln_relu = ln->relu;
  }
  a.gt = nullptr; a.spread = nullptr;
  if (have_shift) {
    if (!aligned16(grad_shifted) && vec != 1) return MLGNN_E_ALIGN;
    a.gt = grad_shifted; a.spread = shift_flag;
  }
  if (want_shift) {
    float* base = workspace + part_floats;               // 16-byte aligned: part_floats is a multiple of 4 when d % 4 == 0
    if (!rowptr) return MLGNN_E_NULL;
    if ((reinterpret_cast<uintptr_t>(base) & 15) == 0 || vec == 1) {
      ShiftArgs sa;
      sa.go = grad_out; sa.lse = aux; sa.rowptr = rowptr; sa.N = (int)N; sa.d = (int)d; sa.lpr_log2 = a.lpr_log2;
      sa.spread = reinterpret_cast<int*>(base); sa.gt = base + 4;
      int err0 = (int)hipMemsetAsync(sa.spread, 0, 16, s);
      if (err0) return err0;
      const int rows_per_block = kWavesPerBlock * (kWave >> a.lpr_log2) * 4;
      int sblk = (int)((N + rows_per_block - 1) / rows_per_block);
      if (sblk > 8192) sblk = 8192;
      if (bf16) {
        if (vec == 8) hipLaunchKernelGGL((softmax_shift_kernel<bf16_t, 8>), dim3(sblk), block, 0, s, sa);
        else hipLaunchKernelGGL((softmax_shift_kernel<bf16_t, 1>), dim3(sblk), block, 0, s, sa);
      } else {
        if (vec == 4) hipLaunchKernelGGL((softmax_shift_kernel<float, 4>), dim3(sblk), block, 0, s, sa);
        else hipLaunchKernelGGL((softmax_shift_kernel<float, 1>), dim3(sblk), block, 0, s, sa);
      }
      a.gt = sa.gt; a.spread = sa.spread;
    }
  }
  a.slot8 = nullptr;
  if (want_slots && rowptr) {
    float* base = workspace + part_floats;
    if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
      SlotArgs sa;
      sa.argmax = argmax; sa.rowptr = rowptr; sa.N = (int)N; sa.d = (int)d;
      sa.spread = reinterpret_cast<int*>(base); sa.slot8 = reinterpret_cast<uint8_t*>(base + 4);
      int err0 = (int)hipMemsetAsync(sa.spread, 0, 16, s);
      if (err0) return err0;
      int64_t sblk = (N * d / 4 + kBlock - 1) / kBlock;
      if (sblk > 4096) sblk = 4096;
      hipLaunchKernelGGL(max_slot_kernel, dim3((unsigned)sblk), block, 0, s, sa);
      a.slot8 = sa.slot8; a.spread = sa.spread;
    }
  }
  const bool lt = learn_t != 0 && ag == A_SOFTMAX;
  // fixed_grid > 0: the launch over the extra chunks of long rows (csrc/hub.hip)
  auto run = [&](const BwdArgs& args, int fixed_grid, auto virt_c) {
    constexpr bool VIRT = decltype(virt_c)::value;
    for_mode_aggr(mode, ag, [&](auto mode_c, auto aggr_c) {
      constexpr int MODE = decltype(mode_c)::value, AGGR = decltype(aggr_c)::value;
      constexpr bool kCanLearn = (AGGR == A_SOFTMAX);
      auto launch = [&](auto t_c, auto vec_c) {
        using T = typename decltype(t_c)::type;
        constexpr int VEC = decltype(vec_c)::value;
        auto go = [&](auto kernel, auto learn_c) {
          // The strided walk is one pass over the rows only while every workgroup is resident (otherwise each
          // generation of workgroups sweeps all graphs again, through a cold L2): size the grid to the occupancy of
          // this instantiation.  (`learn_c` makes the two variants of one (T, VEC, MODE, AGGR) distinct
          // instantiations of this lambda: their kernels have the same function type, and would share the static.)
          static int per_cu = 0;
          (void)learn_c;
          if (per_cu == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlock, 0) != hipSuccess || n < 1) n = 2;
            per_cu = n;
          }
          int g = per_cu * num_cus() / kXcds * kXcds;
          g = g < nblk ? (g < kXcds ? kXcds : g) : nblk;
          if (fixed_grid > 0) g = fixed_grid; else launched_blocks = g;
          hipLaunchKernelGGL(kernel, dim3(g), block, 0, s, args);
        };
        if constexpr (std::is_same<T, float>::value && VEC == 4 && !VIRT) {
          if (ln) {
            if (wide) return;                                  // (refused below: the fold keeps 32-bit rows)
            go(csr_aggregate_bwd_kernel<T, VEC, MODE, AGGR, false, VIRT, true>, IC<2>{});
            return;
          }
        }
        if constexpr (VEC > 1) {
          if (wide) {
            if (kCanLearn && lt) go(csr_aggregate_bwd_kernel<T, VEC, MODE, AGGR, kCanLearn, VIRT, false, true>, IC<3>{});
            else go(csr_aggregate_bwd_kernel<T, VEC, MODE, AGGR, false, VIRT, false, true>, IC<4>{});
            return;
          }
        }
        if (kCanLearn && lt) go(csr_aggregate_bwd_kernel<T, VEC, MODE, AGGR, kCanLearn, VIRT>, BC<true>{});
        else go(csr_aggregate_bwd_kernel<T, VEC, MODE, AGGR, false, VIRT>, BC<false>{});
      };
      if (bf16) { if (vec == 8) launch(TypeTag<bf16_t>{}, IC<8>{}); else launch(TypeTag<bf16_t>{}, IC<1>{}); }
      else { if (vec == 4) launch(TypeTag<float>{}, IC<4>{}); else launch(TypeTag<float>{}, IC<1>{}); }
    });
  };
  // narrow fp32 rows, weighted sum / mean: one lane group per source row (aggregate_short.h)
  static const bool short_on = [] { const char* e = getenv("MLGNN_SHORT_ROWS"); return !(e && e[0] == '0'); }();
  if (short_on && !bf16 && vec == 4 && (mode == M_IDENTITY || mode == M_WEIGHTED) && ag == A_SUM && short_width_ok(d) &&
      rk == 0 && !ln && !add_root && !wide && !grad_efull && (!a.mean || rowptr)) {
    MLGNN_SHORT_DISPATCH(csr_short_bwd_kernel, d, mode == M_WEIGHTED, static_cast<const float*>(grad_out), rowptr_t, col_t,
                         ew_t, rowptr, static_cast<float*>(grad_x), (int)N, a.mean, a.cap);
    launched_blocks = 0;
  } else {
    run(a, 0, BC<false>{});
  }
  int total_blocks = launched_blocks;
  if (split) {
    int err1 = (int)hipGetLastError();
    if (err1) return err1;
    BwdArgs b = a;
    b.cap = kNoCap; b.vrows = hub->vrows; b.vcount = hub->counts;
    b.gx = hub->tmp;                                            // one partial grad_x row per extra chunk
    b.ws = rk > 0 ? workspace + (int64_t)launched_blocks * (rk + 1) * d : nullptr;
    run(b, kHubBlocks, BC<true>{});
    total_blocks += kHubBlocks;
    err1 = (int)hipGetLastError();
    if (err1) return err1;
    err1 = hub_combine_bwd(hub->hubs, hub->counts, grad_x, hub->tmp, (int)d, bf16, s);
    if (err1) return err1;
  }
  int err = (int)hipGetLastError();
  if (err) return err;
  if (rk > 0) {
    launch_reduce_partials(workspace, grad_uv, total_blocks, (rk + 1) * (int)d, s);
    err = (int)hipGetLastError();
  }
  if (ln && !err) {
    launch_reduce_partials(ln->workspace, ln->grad_gamma_beta, launched_blocks, 2 * (int)d, s);
    err = (int)hipGetLastError();
  }
  return err;
}

extern "C" int mlgnn_csr_aggregate_bwd(const void* grad_out, const void* x, const void* out, const float* aux,
                                       const int32_t* argmax,
                                       const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                                       const int32_t* rowptr,
                                       const float* ew_t, const float* eu, const float* ev,
                                       const void* efull, const int32_t* eid_t, const int32_t* geid_t,
                                       void* grad_x, void* grad_efull, float* grad_uv,
                                       float* workspace, int64_t workspace_floats,
                                       int64_t N, int64_t d, int dtype, int msg, int edge_mode, int edge_rank,
                                       int aggr, int learn_t, float t, float p, const float* t_dev,
                                       const float* p_dev, float eps, int add_root, int accumulate_efull,
                                       const mlgnn_hub_t* hub, const void* grad_shifted, const int32_t* shift_flag,
                                       void* stream) {
  return csr_aggregate_bwd_impl(grad_out, x, out, aux, argmax, rowptr_t, col_t, pos_t, rowptr, ew_t, eu, ev, efull, eid_t,
                                geid_t, grad_x, grad_efull, grad_uv, workspace, workspace_floats, N, d, dtype, msg, edge_mode,
                                edge_rank, aggr, learn_t, t, p, t_dev, p_dev, eps, add_root, accumulate_efull, hub,
                                grad_shifted, shift_flag, nullptr, stream);
}

extern "C" int64_t mlgnn_csr_aggregate_bwd_ln_workspace_floats(int64_t N, int64_t d) {
  if (N < 0 || d <= 0 || N > INT32_MAX) return MLGNN_E_SHAPE;
  return (int64_t)grid_for_rows(N) * 2 * d;
}

extern "C" int mlgnn_csr_aggregate_bwd_ln(const void* grad_out, const void* x, const void* out, const float* aux,
                                          const int32_t* argmax,
                                          const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                                          const int32_t* rowptr,
                                          const float* ew_t, const float* eu, const float* ev,
                                          const void* efull, const int32_t* eid_t, const int32_t* geid_t,
                                          void* grad_x, void* grad_efull, float* grad_uv,
                                          float* workspace, int64_t workspace_floats,
                                          int64_t N, int64_t d, int dtype, int msg, int edge_mode, int edge_rank,
                                          int aggr, int learn_t, float t, float p, const float* t_dev,
                                          const float* p_dev, float eps, int add_root, int accumulate_efull,
                                          const mlgnn_hub_t* hub, const void* grad_shifted, const int32_t* shift_flag,
                                          const mlgnn_ln_fold_t* ln, void* stream) {
  if (!ln) return MLGNN_E_NULL;
  return csr_aggregate_bwd_impl(grad_out, x, out, aux, argmax, rowptr_t, col_t, pos_t, rowptr, ew_t, eu, ev, efull, eid_t,
                                geid_t, grad_x, grad_efull, grad_uv, workspace, workspace_floats, N, d, dtype, msg, edge_mode,
                                edge_rank, aggr, learn_t, t, p, t_dev, p_dev, eps, add_root, accumulate_efull, hub,
                                grad_shifted, shift_flag, ln, stream);
}
