// CSR gather-reduce kernels: fused message + neighbour aggregation (forward, by-destination CSR)
// and its backward (by-source CSR).  One wavefront owns one node row; the 64 lanes are split
// into G = 64/LPR groups of LPR lanes, each group streaming whole neighbour rows with 16-byte
// loads (VEC = 4 floats per lane), so one wave-instruction fetches G coalesced rows.  Groups are
// combined with wavefront shuffles at the end of the row -- no atomics, no LDS in the forward.
//
// Reference semantics (paths relative to the reference tree):
//   message    relu(x_j + e_ij) + eps              models/gcn_lib/sparse/torch_vertex.py:94-101
//              x_j * w_ij                          models/gcn_lib/sparse/torch_vertex.py:279-281
//   aggregate  add / mean / max / softmax / power  models/gcn_lib/sparse/torch_message.py:44-85
//
// HBM-bound (0.25-1 FLOP/byte): algorithmic bytes per launch are E*d*4 (neighbour rows)
// + E*4 (col) + (N+1)*4 (rowptr) + E*4 (edge scalar) [+ E*d*4 full edge embedding] + N*d*4 (out).
//
// Built with -fno-honor-nans -fno-honor-infinities (plain v_max/v_min, no canonicalisation):
// "minus infinity" sentinels are the finite kNegBig.
#include "aggregate_common.h"
#include "aggregate_short.h"

namespace mlgnn {

template <typename T, int VEC, int MODE, int AGGR, bool SECOND, bool VIRT = false, bool WIDE = false>
__global__ __launch_bounds__(kBlock) void csr_aggregate_fwd_kernel(const FwdArgs a) {
  const T* X = static_cast<const T*>(a.x);
  const T* EF = static_cast<const T*>(a.efull);
  T* OUT = static_cast<T*>(a.out);
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << a.lpr_log2;
  const int groups = kWave >> a.lpr_log2;
  const int sub = lane >> a.lpr_log2;
  const int cl = lane & (lpr - 1);
  const RowWalk walk = make_chunk_walk(VIRT ? *a.vcount : a.N);
  const Scalars sc = read_scalars(a.t_dev, a.p_dev, a.t, a.p);
  const uint32_t row_bytes = (uint32_t)a.d * (uint32_t)sizeof(T);
  constexpr int RK = rank_of<MODE>();                    // rank of the factored edge term (0: none)
  constexpr int ES = edge_scalars<MODE>();               // scalars per edge read from ew
  constexpr int ESA = ES > 0 ? ES : 1;

  // eps is added once per row instead of once per edge: softmax weights are shift invariant,
  // max and sum commute with the shift (power needs the clamp of m itself and keeps it per edge)
  constexpr bool kLateEps = is_gen<MODE>() && AGGR != A_POWER;
  // non-finite messages are carried to the result (aggregate_common.h) -- except for max: torch_scatter's scatter_max
  // compares (`new > current`), so a NaN message never wins there either; +Inf wins by itself
  constexpr bool kTrack = is_gen<MODE>() && AGGR != A_MAX;

  for (int cbase = 0; cbase < a.d; cbase += lpr * VEC) {
    // lanes past the last channel (d/VEC not a power of two) re-read the last valid chunk and are
    // only masked at the store, so full batches need no per-lane predication at all
    const bool cact = cbase + cl * VEC < a.d;
    const int c0 = min(cbase + cl * VEC, a.d - VEC);
    const uint32_t c_bytes = (uint32_t)c0 * (uint32_t)sizeof(T);
    float eu[VEC][ESA], ev[VEC];                         // eu[i][k] = U[k][c0 + i]
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      ev[i] = 0.f;
#pragma unroll
      for (int k = 0; k < ESA; ++k) eu[i][k] = 0.f;
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

    // the sign of the temperature picks the batch extremum (max / min of the messages); it is uniform for the
    // whole launch, so the row loop is instantiated for both signs and selected once (no branch per batch)
    auto rows = [&](auto tpos_c) {
    constexpr bool TPOS = decltype(tpos_c)::value;
    for (int r = walk.first; r < walk.r_end; r += walk.stride) {
      // a chunk of a long row (csrc/hub.hip) is a partial result: no root term, combined by hub_combine_fwd_kernel
      const int beg = VIRT ? a.vrows[3 * r + 1] : a.rowptr[r];
      const int row_end = VIRT ? a.vrows[3 * r + 2] : a.rowptr[r + 1];
      const int end = VIRT ? row_end : min(row_end, beg + a.cap);
      const bool partial = VIRT || end != row_end;
      const int deg = end - beg;

      // accumulators: SUM/POWER use acc; MAX uses acc (best) + bpos; SOFTMAX uses mx, acc (S), w1, w2
      float acc[VEC], mx[VEC], w1[VEC], w2[VEC];
      int bpos[VEC];
      // FAST (softmax only): no running maximum -- weights 2^(t m) against the fixed reference 0.  Messages are
      // relu outputs of normalised features, so t m stays far inside fp32's exponent range and the per-batch
      // max / rescale bookkeeping (a quarter of the kernel's VALU work) buys nothing; the row is redone with the
      // online recurrence when its sums come out too large or too small to trust (checked below).
      auto scan = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[i] = (AGGR == A_MAX) ? kNegBig : 0.f;
        mx[i] = FAST ? 0.f : kNegBig; w1[i] = 0.f; w2[i] = 0.f; bpos[i] = -1;
      }

      // FAST sums as register pairs from start to end (pairs rebuilt from scalars per batch cost v_movs at every
      // loop boundary): pa = S, p1 = sum w m, p2 = sum w m^2
      constexpr bool kPairs = AGGR == A_SOFTMAX && FAST && VEC % 2 == 0;
      constexpr int NP = VEC % 2 == 0 ? VEC / 2 : 1;
      f32x2 pa[NP], p1[NP], p2[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) { pa[i] = {0.f, 0.f}; p1[i] = {0.f, 0.f}; p2[i] = {0.f, 0.f}; }

      for (int base = beg; base < end; base += kWave) {
        const int cnt = min(kWave, end - base);
        uint32_t my_off = 0;
        int my_eid = 0;
        float my_ew[ESA];
#pragma unroll
        for (int k = 0; k < ESA; ++k) my_ew[k] = 0.f;
        if (lane < cnt) {
          my_off = row_key<WIDE>(a.col[base + lane], row_bytes);
          if constexpr (ES > 0) {
            load_edge_scalars<ES>(my_ew, a.ew, (size_t)(base + lane));
          }
          if (MODE == M_GEN_FULL) my_eid = a.eid[base + lane];
        }

        // one batch = kUnroll neighbours per lane group; FULL batches carry no validity masks
        auto batch = [&](auto full_c, const int k) {
          constexpr bool FULL = decltype(full_c)::value;
          float m[kUnroll][VEC];
          bool valid[kUnroll];
          {
            float xv[kUnroll][VEC], ef[kUnroll][VEC], wa[kUnroll][ESA];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
              const int idx = k + u * groups + sub;
              valid[u] = FULL || (idx < cnt);
              const int src = idx & (kWave - 1);
              const auto off = row_offset<WIDE>((uint32_t)__shfl((int)my_off, src), row_bytes, c_bytes);
#pragma unroll
              for (int q = 0; q < ESA; ++q) wa[u][q] = (ES > 0) ? __shfl(my_ew[q], src) : 0.f;
              const int e0 = (MODE == M_GEN_FULL) ? __shfl(my_eid, src) : 0;
#pragma unroll
              for (int i = 0; i < VEC; ++i) { xv[u][i] = 0.f; ef[u][i] = 0.f; }
              if (FULL || valid[u]) {
                load_row<T, VEC>(xv[u], X, off);
                if (MODE == M_GEN_FULL) load_t<T, VEC>(ef[u], EF + (size_t)e0 * a.d + c0);
              }
            }
            // lane-group chunks of a partial batch that lie past the row's last edge are skipped as a whole
            // (wave-uniform branch): a row of 16 edges ends in a partial batch more often than not
            const int live = FULL ? kUnroll : (cnt - k + groups - 1) >> (6 - a.lpr_log2);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
              if (!FULL && u >= live) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) m[u][i] = 0.f;
                continue;
              }
              float zu[VEC];
              row_messages<MODE, VEC, ESA, kTrack, !kLateEps>(m[u], zu, xv[u], wa[u], eu, ev, ef[u], a.eps);
              const bool ok = FULL || valid[u];
              if constexpr (AGGR == A_SUM) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += ok ? m[u][i] : 0.f;
              } else if constexpr (AGGR == A_MAX) {
                const int pos = base + k + u * groups + sub;
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                  if (ok && m[u][i] > acc[i]) { acc[i] = m[u][i]; bpos[i] = pos; }
              } else if constexpr (AGGR == A_SOFTMAX && FAST) {
                if constexpr (kPairs) {
#pragma unroll
                  for (int i = 0; i < VEC; i += 2) {
                    const f32x2 m2 = {m[u][i], m[u][i + 1]}, t2 = {sc.t_log2e, sc.t_log2e};
                    const f32x2 tm = m2 * t2;
                    f32x2 pe = {fast_exp2(tm.x), fast_exp2(tm.y)};
                    if (!FULL) { pe.x = ok ? pe.x : 0.f; pe.y = ok ? pe.y : 0.f; }
                    pa[i / 2] = pa[i / 2] + pe;
                    p1[i / 2] = __builtin_elementwise_fma(pe, m2, p1[i / 2]);
                    if (SECOND) p2[i / 2] = __builtin_elementwise_fma(pe * m2, m2, p2[i / 2]);
                  }
                } else {
#pragma unroll
                  for (int i = 0; i < VEC; ++i) {
                    float pe = fast_exp2(sc.t_log2e * m[u][i]);
                    if (!FULL) pe = ok ? pe : 0.f;
                    acc[i] += pe;
                    w1[i] = fmaf(pe, m[u][i], w1[i]);
                    if (SECOND) w2[i] = fmaf(pe * m[u][i], m[u][i], w2[i]);
                  }
                }
              } else if constexpr (AGGR == A_POWER) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                  const float mc = clamp_nan(m[u][i], kPowLo, kPowHi);     // torch.clamp carries a NaN message
                  const float l2 = fast_log2(mc);
                  const float pw = fast_exp2(sc.p * l2);
                  acc[i] += ok ? pw : 0.f;
                  if (SECOND) w2[i] += ok ? pw * l2 * kLn2 : 0.f;
                }
              }
            }
          }

          if constexpr (AGGR == A_SOFTMAX && !FAST) {
            // online softmax, one rescale per batch of kUnroll neighbours; units: log2.  t*m is monotone in m:
            // the batch extremum of m (max for t >= 0, min for t < 0) gives the extremum of t*m.
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              if (!FULL && !valid[0]) break;
              float ext = m[0][i];
#pragma unroll
              for (int u = 1; u < kUnroll; ++u) {
                const float cand = TPOS ? fmaxf(ext, m[u][i]) : fminf(ext, m[u][i]);
                ext = (FULL || valid[u]) ? cand : ext;
              }
              const float zmax = fmaxf(mx[i], sc.t_log2e * ext);
              const float rs = fast_exp2(mx[i] - zmax);     // 0 on the first batch (mx = kNegBig)
              float s = acc[i] * rs, s1 = w1[i] * rs, s2 = SECOND ? w2[i] * rs : 0.f;
#pragma unroll
              for (int u = 0; u < kUnroll; ++u) {
                float pe = fast_exp2(fmaf(sc.t_log2e, m[u][i], -zmax));
                if (!FULL) pe = valid[u] ? pe : 0.f;
                s += pe;
                s1 = fmaf(pe, m[u][i], s1);
                if (SECOND) s2 = fmaf(pe * m[u][i], m[u][i], s2);
              }
              acc[i] = s; w1[i] = s1; w2[i] = s2; mx[i] = zmax;
            }
          }
        };

        const int step = groups * kUnroll;
        int k = 0;
        for (; k + step <= cnt; k += step) batch(BC<true>{}, k);
        if (k < cnt) batch(BC<false>{}, k);
      }
      if constexpr (kPairs) {
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
          acc[i] = pa[i / 2].x; acc[i + 1] = pa[i / 2].y;
          w1[i] = p1[i / 2].x; w1[i + 1] = p1[i / 2].y;
          w2[i] = p2[i / 2].x; w2[i + 1] = p2[i / 2].y;
        }
      }

      // ---- combine the lane groups (xor-shuffle over the group bits) ----
      for (int off = lpr; off < kWave; off <<= 1) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if constexpr (AGGR == A_SUM) {
            acc[i] += __shfl_xor(acc[i], off);
          } else if constexpr (AGGR == A_POWER) {
            acc[i] += __shfl_xor(acc[i], off);
            if (SECOND) w2[i] += __shfl_xor(w2[i], off);
          } else if constexpr (AGGR == A_MAX) {
            const float ov = __shfl_xor(acc[i], off);
            const int op = __shfl_xor(bpos[i], off);
            // larger value wins; on a tie the earlier edge (torch_scatter CPU keeps the first)
            const bool take = (op >= 0) && (bpos[i] < 0 || ov > acc[i] || (ov == acc[i] && op < bpos[i]));
            if (take) { acc[i] = ov; bpos[i] = op; }
          } else if constexpr (FAST) {   // SOFTMAX against the fixed reference: plain sums
            acc[i] += __shfl_xor(acc[i], off);
            w1[i] += __shfl_xor(w1[i], off);
            if (SECOND) w2[i] += __shfl_xor(w2[i], off);
          } else {  // SOFTMAX: a group that saw no edge has (mx, S, W) = (kNegBig, 0, 0)
            const float om = __shfl_xor(mx[i], off);
            const float os = __shfl_xor(acc[i], off);
            const float o1 = __shfl_xor(w1[i], off);
            const float nm = fmaxf(mx[i], om);
            const float sa = fast_exp2(mx[i] - nm), sb = fast_exp2(om - nm);
            acc[i] = acc[i] * sa + os * sb;
            w1[i] = w1[i] * sa + o1 * sb;
            if (SECOND) { const float o2 = __shfl_xor(w2[i], off); w2[i] = w2[i] * sa + o2 * sb; }
            mx[i] = nm;
          }
        }
      }

      };   // scan
      if constexpr (AGGR == A_SOFTMAX) {
        scan(BC<true>{});
        bool bad = false;
#pragma unroll
        for (int i = 0; i < VEC; ++i) bad |= !(acc[i] > 1.0e-30f && acc[i] < 1.0e30f);
        if (deg > 0 && __builtin_amdgcn_ballot_w64(bad) != 0) scan(BC<false>{});      // wave-uniform: redo the row
      } else {
        scan(BC<false>{});
      }

      // ---- epilogue: group 0 writes the row ----
      if (sub == 0 && cact) {
        float o[VEC], ax[VEC], ax2[VEC];
        int am[VEC];
        const float inv = __builtin_amdgcn_rcpf((float)max(deg, 1));
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          ax[i] = 0.f; ax2[i] = 0.f; am[i] = -1;
          if constexpr (AGGR == A_SUM) {
            const float tot = kLateEps ? fmaf((float)deg, a.eps, acc[i]) : acc[i];
            o[i] = a.mean ? tot * inv : tot;
          } else if constexpr (AGGR == A_MAX) {
            o[i] = (bpos[i] >= 0) ? acc[i] + (kLateEps ? a.eps : 0.f) : 0.f;
            am[i] = acc[i] > 0.f ? bpos[i] : -1;     // a winner on relu's flat side carries no gradient: not named
          } else if constexpr (AGGR == A_SOFTMAX) {
            if (deg > 0) {
              const float rs = __builtin_amdgcn_rcpf(acc[i]);
              const float o0 = w1[i] * rs;                       // sum_e w_e relu(z_e)
              o[i] = o0 + a.eps;
              // lse of t*(relu(z)+eps), sum_e w_e (relu(z_e)+eps)^2
              ax[i] = mx[i] + fast_log2(acc[i]) + sc.t_log2e * a.eps;
              ax2[i] = fmaf(a.eps, fmaf(2.f, o0, a.eps), w2[i] * rs);
            } else { o[i] = 0.f; }
          } else {  // POWER
            const float mu = acc[i] * inv;
            const float muc = clamp_nan(mu, kPowLo, kPowHi);      // the outer clamp carries NaN too (torch_message.py:72)
            o[i] = fast_exp2(fast_log2(muc) * __builtin_amdgcn_rcpf(sc.p));
            ax[i] = mu;
            ax2[i] = w2[i] * inv;
          }
        }
        const size_t off = (size_t)r * a.d + c0;
        if (a.add_root && !partial) { // h = x_i + m_i (GENConv.forward, torch_vertex.py:89) in the same pass
          float xr[VEC];
          load_t<T, VEC>(xr, X + off);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] += xr[i];
        }
        if constexpr (VIRT) store_vec<VEC>(reinterpret_cast<float*>(a.out) + off, o);   // chunk partials stay fp32 (hub.hip)
        else store_t_stream<T, VEC>(OUT + off, o);
        if (AGGR == A_MAX && a.argmax) store_vec<VEC>(a.argmax + off, am);
        if ((AGGR == A_SOFTMAX || AGGR == A_POWER) && a.aux) store_t_stream<float, VEC>(a.aux + off, ax);
        if (SECOND && a.aux2) store_t_stream<float, VEC>(a.aux2 + off, ax2);
        if (a.rowmax) {   // max |row| of the result for the consumer GEMM's per-row scaling.  Only requested when
                          // d == lpr * VEC: every lane of group 0 is in this branch, so the shuffles are defined
          float om = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) om = fmaxf(om, fabsf(o[i]));
          for (int sh = 1; sh < lpr; sh <<= 1) om = fmaxf(om, __shfl_xor(om, sh));
          if (cl == 0) a.rowmax[r] = om;
        }
      }
    }
    };
    if constexpr (AGGR == A_SOFTMAX) {
      if (sc.t_log2e >= 0.f) rows(BC<true>{}); else rows(BC<false>{});
    } else {
      rows(BC<true>{});
    }
  }
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" int mlgnn_version(void) { return MLGNN_ABI_VERSION; }

extern "C" int64_t mlgnn_csr_aggregate_bwd_workspace_floats(int64_t N, int64_t d, int dtype, int edge_rank,
                                                            int aggr, int learn_t) {
  if (N < 0 || d < 0 || edge_rank < 0 || edge_rank > 8) return MLGNN_E_SHAPE;
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  // per-workgroup partials of the factored edge term + the softmax shift / max winner-slot buffers (aggregate_bwd.hip)
  int64_t n = edge_rank > 0 ? (int64_t)(grid_for_rows(N) + kHubBlocks) * (edge_rank + 1) * d : 0;
  if (aggr == MLGNN_AGGR_SOFTMAX && !learn_t)
    n += 4 + (N * d * (dtype == MLGNN_DTYPE_BF16 ? 2 : 4) + 3) / 4;
  if (aggr == MLGNN_AGGR_MAX && d % 4 == 0) n += 4 + (N * d + 3) / 4;      // flag + one-byte winner slots
  return n;
}

extern "C" int mlgnn_csr_aggregate_fwd(const void* x, const int32_t* rowptr, const int32_t* col,
                                       const float* ew, const float* eu, const float* ev,
                                       const void* efull, const int32_t* eid,
                                       void* out, float* aux, float* aux2, int32_t* argmax, float* row_max,
                                       int64_t N, int64_t d, int dtype, int msg, int edge_mode, int edge_rank,
                                       int aggr, float t, float p, const float* t_dev, const float* p_dev,
                                       float eps, int add_root, const mlgnn_hub_t* hub, void* stream) {
  if (dtype != MLGNN_DTYPE_F32 && dtype != MLGNN_DTYPE_BF16) return MLGNN_E_DTYPE;
  if (N < 0 || d <= 0 || N > INT32_MAX || d > INT32_MAX) return MLGNN_E_SHAPE;
  if (hub && hub->cap > 0) {
    if (!hub->vrows || !hub->hubs || !hub->counts || !hub->tmp) return MLGNN_E_NULL;
    if (hub->capacity < 1 || hub->tmp_bytes < mlgnn_hub_scratch_bytes(hub->capacity, d)) return MLGNN_E_WORKSPACE;
    if ((int64_t)hub->capacity * d * 4 >= (int64_t)1 << 32) return MLGNN_E_SHAPE;
  }
  const int mode = pick_mode(msg, edge_mode, edge_rank);
  const int ag = pick_aggr(aggr);
  if (mode < 0 || ag < 0) return MLGNN_E_MODE;
  if (!is_gen_mode(mode) && ag != A_SUM) return MLGNN_E_MODE;
  if (N == 0) return 0;
  if (!x || !rowptr || !out) return MLGNN_E_NULL;   // col may be NULL iff the graph has no edge
  if ((mode == M_WEIGHTED || rank_of_mode(mode) > 0) && !ew && col) return MLGNN_E_NULL;
  if (rank_of_mode(mode) > 0 && (!eu || !ev)) return MLGNN_E_NULL;
  if (mode == M_GEN_FULL && col && (!efull || !eid)) return MLGNN_E_NULL;
  if (ag == A_POWER && !p_dev && !(p != 0.0f)) return MLGNN_E_MODE;

  FwdArgs a;
  a.x = x; a.rowptr = rowptr; a.col = col; a.ew = ew; a.eu = eu; a.ev = ev;
  a.efull = efull; a.eid = eid; a.out = out; a.aux = aux; a.aux2 = aux2;
  a.argmax = argmax; a.N = (int)N; a.d = (int)d; a.mean = (aggr == MLGNN_AGGR_MEAN);
  a.t = t; a.p = p; a.eps = eps; a.t_dev = t_dev; a.p_dev = p_dev; a.add_root = add_root;
  const bool split = hub && hub->cap > 0 && col;
  a.cap = split ? hub->cap : kNoCap; a.vrows = nullptr; a.vcount = nullptr;

  const bool al = aligned16(x) && aligned16(out) && (!efull || aligned16(efull)) &&
                  (!aux || aligned16(aux)) && (!aux2 || aligned16(aux2)) && (!argmax || aligned16(argmax)) &&
                  (!eu || aligned16(eu)) && (!ev || aligned16(ev));
  // the per-edge scalar table is read with 4 * rank byte loads (rank 2: float2, 4 / 8: float4) on every path
  {
    const int es = mode == M_WEIGHTED ? 1 : rank_of_mode(mode);
    const uintptr_t need = es >= 4 ? 16 : 4 * (uintptr_t)(es > 0 ? es : 1);
    if (ew && (reinterpret_cast<uintptr_t>(ew) % need) != 0) return MLGNN_E_ALIGN;
  }
  const bool bf16 = dtype == MLGNN_DTYPE_BF16;
  // channels per lane: 16-byte accesses (4 x fp32 / 8 x bf16) when the width allows, scalar otherwise
  const int vec = bf16 ? ((d % 8 == 0 && al) ? 8 : 1) : ((d % 4 == 0 && al) ? 4 : 1);
  const dim3 grid(grid_for_chunks(N, kFwdRowsPerWave)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  const bool second = (aux2 != nullptr) && (ag == A_SOFTMAX || ag == A_POWER);
  a.lpr_log2 = lanes_per_row_log2(d, vec);
  a.rowmax = row_max;
  // [N,d] of 4 GiB and more: 64-bit row addresses (16-byte path only: the scalar fallback is for odd widths of small inputs)
  const bool wide = needs_wide_rows(N, d) || force_wide_rows();
  if (wide && vec == 1 && needs_wide_rows(N, d)) return MLGNN_E_SHAPE;
  if (row_max && d != ((int64_t)vec << a.lpr_log2)) return MLGNN_E_SHAPE;    // row max: one chunk, no shadow lanes
  auto run = [&](const FwdArgs& args, const dim3 g, auto virt_c) {
    constexpr bool VIRT = decltype(virt_c)::value;
    for_mode_aggr(mode, ag, [&](auto mode_c, auto aggr_c) {
      constexpr int MODE = decltype(mode_c)::value, AGGR = decltype(aggr_c)::value;
      constexpr bool kHasSecond = (AGGR == A_SOFTMAX || AGGR == A_POWER);
      auto launch = [&](auto t_c, auto vec_c) {
        using T = typename decltype(t_c)::type;
        constexpr int VEC = decltype(vec_c)::value;
        if constexpr (VEC > 1) {
          if (wide) {
            if (kHasSecond && second) hipLaunchKernelGGL((csr_aggregate_fwd_kernel<T, VEC, MODE, AGGR, kHasSecond, VIRT, true>), g, block, 0, s, args);
            else hipLaunchKernelGGL((csr_aggregate_fwd_kernel<T, VEC, MODE, AGGR, false, VIRT, true>), g, block, 0, s, args);
            return;
          }
        }
        if (kHasSecond && second) hipLaunchKernelGGL((csr_aggregate_fwd_kernel<T, VEC, MODE, AGGR, kHasSecond, VIRT>), g, block, 0, s, args);
        else hipLaunchKernelGGL((csr_aggregate_fwd_kernel<T, VEC, MODE, AGGR, false, VIRT>), g, block, 0, s, args);
      };
      if (bf16) { if (vec == 8) launch(TypeTag<bf16_t>{}, IC<8>{}); else launch(TypeTag<bf16_t>{}, IC<1>{}); }
      else { if (vec == 4) launch(TypeTag<float>{}, IC<4>{}); else launch(TypeTag<float>{}, IC<1>{}); }
    });
  };
  // narrow fp32 rows, weighted sum / mean (the SAGE layers of the shipped configs): one lane group per row
  // (aggregate_short.h); same clamp-to-cap contract, so the long-row launches below follow either kernel
  static const bool short_on = [] { const char* e = getenv("MLGNN_SHORT_ROWS"); return !(e && e[0] == '0'); }();
  if (short_on && !bf16 && vec == 4 && (mode == M_IDENTITY || mode == M_WEIGHTED) && ag == A_SUM && short_width_ok(d) &&
      !aux && !aux2 && !argmax && !row_max && !add_root && !wide) {
    MLGNN_SHORT_DISPATCH(csr_short_fwd_kernel, d, mode == M_WEIGHTED, static_cast<const float*>(x), rowptr, col, ew,
                         static_cast<float*>(out), (int)N, a.mean, a.cap);
  } else {
    run(a, grid, BC<false>{});
  }
  int err = (int)hipGetLastError();
  if (err || !split) return err;

  // long rows: the chunks behind the first `cap` edges of every row, then the fixed-order combine (csrc/hub.hip)
  const size_t rows = (size_t)hub->capacity;
  unsigned char* tmp = static_cast<unsigned char*>(hub->tmp);
  auto carve = [&](size_t bytes) { unsigned char* q = tmp; tmp += (bytes + 63) & ~(size_t)63; return q; };
  void* out_v = carve(rows * d * 4);                               // chunk partials are fp32 whatever the storage type
  float* aux_v = reinterpret_cast<float*>(carve(rows * d * 4));
  float* aux2_v = reinterpret_cast<float*>(carve(rows * d * 4));
  int* arg_v = reinterpret_cast<int*>(carve(rows * d * 4));
  FwdArgs b = a;
  b.cap = kNoCap; b.vrows = hub->vrows; b.vcount = hub->counts;
  b.out = out_v; b.aux = aux ? aux_v : nullptr; b.aux2 = aux2 ? aux2_v : nullptr; b.argmax = argmax ? arg_v : nullptr;
  b.rowmax = nullptr; b.add_root = 0;
  run(b, dim3(256), BC<true>{});
  err = (int)hipGetLastError();
  if (err) return err;
  HubFwdArgs h;
  h.hubs = hub->hubs; h.vrows = hub->vrows; h.counts = hub->counts; h.rowptr = rowptr;
  h.out = out; h.aux = aux; h.aux2 = aux2; h.argmax = argmax; h.rowmax = row_max; h.x = x;
  h.out_v = out_v; h.aux_v = aux_v; h.aux2_v = aux2_v; h.argmax_v = arg_v;
  h.p_dev = p_dev; h.p = p; h.d = (int)d; h.cap = hub->cap; h.aggr = ag; h.mean = a.mean; h.add_root = add_root;
  h.second = second ? 1 : 0;
  if ((ag == A_SOFTMAX || ag == A_POWER) && !aux) return MLGNN_E_NULL;      // the combine needs the chunks' lse / means
  return hub_combine_fwd(h, bf16, s);
}

