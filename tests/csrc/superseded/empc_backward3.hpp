// empc_backward3.hpp -- HOT-B kernel body, matrix-core form: one wavefront per trajectory, the three dense products of a
// node (W = Vxx' [Fx Fu], Q = H + [Fx Fu]^T W with Qx/Qu as an extra column, Vxx = Qxx - Qxu K) on v_mfma_f64_16x16x4_f64.
// Operand layout (tools/probes/mfma_f64_layout.hip, verified on gfx950): A[i = lane % 16][k = lane / 16],
// B[k = lane / 16][j = lane % 16], D[i = 4 r + lane / 16][j = lane % 16] for the four result registers r.
// The FP64 matrix rate equals the vector rate on this chip; what the matrix cores buy here is operand traffic: one LDS
// read feeds a 16 x 16 x 4 block (1024 FMAs) instead of one FMA, and ~1200 vector instructions per node become 52 MFMAs.
// Everything else (prologue, LLT and gain solves, symmetrisation, gap terms, regularisation retry loop) is the code of
// empc_backward2.hpp, which stays as the vector form (emulator cross-check, NL = 256 tiling).
//
#pragma once
// TEST INFRASTRUCTURE (superseded kernel form, kept as a cross-check of the shipped one through the CPU lane emulator;
// not compiled into libempc.so)
#include "empc_backward2.hpp"

namespace empc {

#ifndef BWD_STAMP
#define BWD_STAMP(i) \
  do {               \
  } while (0)
#endif

template <class DM>
struct Bwd3Smem {
  static constexpr int n = DM::NDX, m = DM::NU, nm = n + m;
  static constexpr int WLD = ((nm + 1 + 15) / 16) * 16;  // W rows hold nm columns + the Vx' column, padded to whole tiles
  static constexpr int QLD = nm + 1;
  static constexpr int OFF_REC = 0;
  static constexpr int OFF_V = (DM::REC + 63) / 64 * 64;  // the record is staged in whole 64-double rows
  static constexpr int OFF_VX = OFF_V + n * n;
  static constexpr int OFF_W = OFF_VX + n;        // n x nm
  static constexpr int OFF_Q = OFF_W + n * WLD;   // nm x nm (leading dimension QLD)
  static constexpr int OFF_QV = OFF_Q + nm * QLD; // nm
  static constexpr int OFF_K = OFF_QV + nm;       // m x n
  static constexpr int OFF_KF = OFF_K + m * n;    // k (m), Quuk (m)
  static constexpr int OFF_RED = OFF_KF + 2 * m;  // 4 x 32 partial sums
  static constexpr int OFF_FLAG = OFF_RED + 128;
  static constexpr int OFF_DUMP = OFF_FLAG + 6;   // write-only word: accumulator entries outside a matrix land here
  static constexpr int OFF_PRO = OFF_FLAG + 8;    // prologue reductions: 3 x 256
  static constexpr int SIZE = (OFF_PRO + 3 * 256 + 1) / 2 * 2;
};

// Exec concept additions used here: ex.sync() is a barrier over all NL lanes, ex.any(f) an OR-reduction + barrier.
template <class DM, class Exec>
EMPC_HD void backward_traj3(Exec& ex, const DevBuffers& D, int b, double* smem) {
  typedef Bwd3Smem<DM> SM;
  constexpr int NL = 64;
  constexpr int n = DM::NDX, m = DM::NU, nm = n + m, REC = DM::REC;
  constexpr int CW = 32;                    // column groups of 32 lanes (nm <= 32) ...
  static_assert(nm <= 64, "state + control dimension too large for the column tiling");
  constexpr int CWE = (nm <= 32) ? 32 : 64; // ... or 64 lanes for the largest robots
  constexpr int NG = NL / CWE;              // row groups
  constexpr int PRE = (REC + NL - 1) / NL;  // prefetch registers per lane
  constexpr int WLD = SM::WLD, QLD = SM::QLD;
  constexpr int MTN = (n + 15) / 16;        // 16-row tiles over n
  constexpr int MTQ = (nm + 15) / 16;       // 16-row tiles over n + m
  constexpr int NTQ = (nm + 1 + 15) / 16;   // 16-column tiles over n + m + 1 (the extra column carries Vx' -> Qx, Qu)
  constexpr int KSN = (n + 3) / 4;          // k steps over n
  constexpr int KSM = (m + 3) / 4;          // k steps over m
  (void)CW;
  TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const int T = D.T;
  double* rec = smem + SM::OFF_REC;
  double* V = smem + SM::OFF_V;
  double* vx = smem + SM::OFF_VX;
  double* W = smem + SM::OFF_W;
  double* Q = smem + SM::OFF_Q;
  double* qv = smem + SM::OFF_QV;
  double* Ks = smem + SM::OFF_K;
  double* kf = smem + SM::OFF_KF;
  double* red = smem + SM::OFF_RED;
  double* flag = smem + SM::OFF_FLAG;
  double* pro = smem + SM::OFF_PRO;
  static_assert(NL <= 256, "prologue reduction area holds 256 lanes");
  const double* tape = D.tape + (size_t)b * (T + 1) * REC;

  // ---- prologue: cost, gap norms, feasibility --------------------------------------------------------------------
  double cost = st.cost, gapnorm = st.gapnorm;
  int is_feasible = st.is_feasible;
  if (st.need_lin) {
    ex.each([&](int lane, int sl) {
      double c = 0, mx = 0, l1 = 0;
      for (int t = lane; t <= T; t += NL) {
        const double* r = tape + (size_t)t * REC;
        c += r[DM::OFF_COST];
        for (int i = 0; i < n; ++i) {
          const double g = fabs(r[DM::OFF_GAP + i]);
          mx = fmax(mx, g);
          l1 += g;
        }
      }
      pro[lane] = c;
      pro[256 + lane] = mx;
      pro[512 + lane] = l1;
    });
    ex.sync();
    double tot_c = 0, tot_mx = 0, tot_l1 = 0;
    for (int i = 0; i < NL; ++i) {
      tot_c += pro[i];
      tot_mx = fmax(tot_mx, pro[256 + i]);
      tot_l1 += pro[512 + i];
    }
    ex.sync();
    cost = tot_c;
    if (!is_feasible) is_feasible = (tot_mx < D.gaptol) ? 1 : 0;
    gapnorm = (P.prm.gap_norm == EMPC_GAP_L1) ? tot_l1 : tot_mx;
  }
  const bool infeas = !is_feasible;

  // where each accumulator entry of the Q stage starts from: H = [[Lxx Lxu | Lx], [. Luu | Lu]] inside the record
  int hidx[Exec::SLOTS][(DM::NDX + DM::NU + 15) / 16][(DM::NDX + DM::NU + 1 + 15) / 16][4];
  ex.each([&](int lane, int sl) {
    const int lj = lane % 16, lq = lane / 16;
#pragma unroll
    for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * mt + 4 * r + lq, j = 16 * nt + lj;
          int idx = -1;
          if (i < n && j < nm)
            idx = DM::OFF_HX + i * nm + j;
          else if (i < n && j == nm)
            idx = DM::OFF_LX + i;
          else if (i >= n && i < nm && j >= n && j < nm)
            idx = DM::OFF_LUU + (i - n) * m + (j - n);
          else if (i >= n && i < nm && j == nm)
            idx = DM::OFF_LU + (i - n);
          hidx[sl][mt][nt][r] = idx;
        }
  });
  double xreg = st.xreg, ureg = st.ureg;
  double dg_u = 0, dq_u = 0, dg_f = 0, dq_f = 0, qu2 = 0;
  bool failed_final = false;
  while (true) {
    bool fail = false;
    dg_u = dq_u = dg_f = dq_f = qu2 = 0;
    // ---- terminal node ---------------------------------------------------------------------------------------
    {
      const double* r = tape + (size_t)T * REC;
      ex.each([&](int lane, int sl) {
        for (int i = lane; i < n * n; i += NL) V[i] = r[DM::OFF_LXX + (i / n) * nm + (i % n)] + (((i / n) == (i % n)) ? xreg : 0.0);
        if (lane < n) {
          vx[lane] = r[DM::OFF_LX + lane];
          qv[lane] = r[DM::OFF_GAP + lane];  // gap of node T staged in qv
        }
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane >= n) return;
        double a_ = 0;
        if (infeas)
          for (int j = 0; j < n; ++j) a_ += V[lane * n + j] * qv[j];
        const double nv = vx[lane] + (infeas ? a_ : 0.0);
        D.Vf[((size_t)b * (T + 1) + T) * n + lane] = a_;
        D.Vx[((size_t)b * (T + 1) + T) * n + lane] = nv;
        red[lane] = infeas ? nv * qv[lane] : 0.0;
        red[32 + lane] = infeas ? qv[lane] * a_ : 0.0;
        W[lane] = nv;
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane < n) vx[lane] = W[lane];
      });
      for (int i = 0; i < n; ++i) {
        dg_f -= red[i];
        dq_f += red[32 + i];
      }
      ex.sync();
    }
    // first record of the sweep
    double pre[Exec::SLOTS][PRE];
    ex.each([&](int lane, int sl) {
      const double* r = tape + (size_t)(T - 1) * REC;
#pragma unroll
      for (int q = 0; q < PRE; ++q) pre[sl][q] = r[lane + q * NL];  // whole rows: the tape has one row of slack
    });
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long bst[16];
    for (int i = 0; i < 16; ++i) bst[i] = 0;
    bst[15] = __builtin_readcyclecounter();
#endif
    for (int t = T - 1; t >= 0; --t) {
      BWD_STAMP(7);
      ex.each([&](int lane, int sl) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) rec[lane + q * NL] = pre[sl][q];
        if (t > 0) {
          const double* r = tape + (size_t)(t - 1) * REC;
#pragma unroll
          for (int q = 0; q < PRE; ++q) pre[sl][q] = r[lane + q * NL];
        }
      });
      ex.sync();
      BWD_STAMP(0);
      // W = V' A, A = [Fx Fu]: MTN x NTQ tiles, KSN steps; out-of-range operand entries are zeros
      double accW[Exec::SLOTS][MTN][NTQ][4];
      ex.each([&](int lane, int sl) {
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) accW[sl][mt][nt][r] = 0.0;
      });
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks) {
        double aop[Exec::SLOTS][MTN], bop[Exec::SLOTS][NTQ];
        ex.each([&](int lane, int sl) {
          const int li = lane % 16, lk = 4 * ks + lane / 16;
          const int kc = (lk < n) ? lk : 0;
#pragma unroll
          for (int mt = 0; mt < MTN; ++mt) {
            const int i = 16 * mt + li;
            const double v = V[((i < n) ? i : 0) * n + kc];
            aop[sl][mt] = (i < n && lk < n) ? v : 0.0;
          }
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt) {
            const int j = 16 * nt + li;
            const double v = rec[DM::OFF_A + kc * nm + ((j < nm) ? j : 0)];
            bop[sl][nt] = (j < nm && lk < n) ? v : 0.0;
          }
        });
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt) ex.mfma(aop, mt, bop, nt, accW, mt, nt);
      }
      ex.each([&](int lane, int sl) {
        const int lj = lane % 16, lq = lane / 16;
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * mt + 4 * r + lq, j = 16 * nt + lj;
              smem[(i < n && j < nm) ? SM::OFF_W + i * WLD + j : SM::OFF_DUMP] = accW[sl][mt][nt][r];  // branch-free store
            }
        if (lane < n) W[lane * WLD + nm] = vx[lane];  // extra column: Vx' (gives Qx, Qu in column nm of Q)
      });
      ex.sync();
      BWD_STAMP(1);
      // Q = H + A^T [W | Vx']: MTQ x NTQ tiles, KSN steps; accumulators start from H = [[Lxx Lxu | Lx], [. Luu | Lu]]
      double accQ[Exec::SLOTS][MTQ][NTQ][4];
      ex.each([&](int lane, int sl) {
#pragma unroll
        for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int idx = hidx[sl][mt][nt][r];  // position of H's entry in the record, -1 outside H
              const double v = rec[idx < 0 ? 0 : idx];
              accQ[sl][mt][nt][r] = (idx < 0) ? 0.0 : v;
            }
      });
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks) {
        double aop[Exec::SLOTS][MTQ], bop[Exec::SLOTS][NTQ];
        ex.each([&](int lane, int sl) {
          const int li = lane % 16, lk = 4 * ks + lane / 16;
          const int kc = (lk < n) ? lk : 0;
#pragma unroll
          for (int mt = 0; mt < MTQ; ++mt) {
            const int i = 16 * mt + li;  // (A^T)[i][k] = A[k][i]
            const double v = rec[DM::OFF_A + kc * nm + ((i < nm) ? i : 0)];
            aop[sl][mt] = (i < nm && lk < n) ? v : 0.0;
          }
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt) {
            const int j = 16 * nt + li;
            const double v = W[kc * WLD + ((j <= nm) ? j : 0)];
            bop[sl][nt] = (j <= nm && lk < n) ? v : 0.0;
          }
        });
#pragma unroll
        for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt) ex.mfma(aop, mt, bop, nt, accQ, mt, nt);
      }
      ex.each([&](int lane, int sl) {
        const int lj = lane % 16, lq = lane / 16;
#pragma unroll
        for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * mt + 4 * r + lq, j = 16 * nt + lj;
              int dst = SM::OFF_DUMP;
              if (i < nm && j < nm && !(i >= n && j < n)) dst = SM::OFF_Q + i * QLD + j;
              if (i < nm && j == nm) dst = SM::OFF_QV + i;
              smem[dst] = accQ[sl][mt][nt][r];
            }
      });
      ex.sync();
      BWD_STAMP(2);
      // computeGains in the first wavefront: LLT(Quu + ureg I); K = Quu^-1 Qxu^T ; k = Quu^-1 Qu ; Quuk
      ex.each([&](int lane, int sl) {
        if (lane >= 64) return;
        double Lq[m * (m + 1) / 2];
#pragma unroll
        for (int i = 0; i < m; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) Lq[i * (i + 1) / 2 + j] = Q[(n + i) * QLD + n + j] + ((i == j) ? ureg : 0.0);
        const bool pd = chol_packed<m>(Lq);
        if (lane == 0) flag[0] = pd ? 0.0 : 1.0;
        if (lane <= n) {
          double rhs[m];
#pragma unroll
          for (int i = 0; i < m; ++i) rhs[i] = (lane < n) ? Q[lane * QLD + n + i] : qv[n + i];
          chol_solve_packed<m>(Lq, rhs);
          if (lane < n) {
#pragma unroll
            for (int i = 0; i < m; ++i) Ks[i * n + lane] = rhs[i];
          } else {
#pragma unroll
            for (int i = 0; i < m; ++i) kf[i] = rhs[i];
#pragma unroll
            for (int i = 0; i < m; ++i) {
              double a_ = 0;
#pragma unroll
              for (int j = 0; j < m; ++j) a_ += Q[(n + (i > j ? i : j)) * QLD + n + (i > j ? j : i)] * rhs[j];
              kf[m + i] = a_ + ureg * rhs[i];
            }
          }
        }
      });
      ex.sync();
      if (flag[0] != 0.0) {
        fail = true;
        break;
      }
      for (int i = 0; i < m; ++i) {
        dg_u += qv[n + i] * kf[i];
        dq_u -= kf[i] * kf[m + i];
        qu2 += qv[n + i] * qv[n + i];
      }
      BWD_STAMP(3);
      // gains out; Vx = Qx + K^T Quuk - 2 K^T Qu (into red); Vxx = Qxx - Qxu K on the matrix cores: the Qxx tiles are
      // still in the accumulators of the Q stage, A operand = -Qxu, B operand = K
      ex.each([&](int lane, int sl) {
        double* Kg = D.K + ((size_t)b * T + t) * m * n;
        for (int i = lane; i < m * n; i += NL) Kg[i] = Ks[i];
        if (lane < m) D.kff[((size_t)b * T + t) * m + lane] = kf[lane];
        if (lane < n) {
          double a_ = qv[lane];
#pragma unroll
          for (int l = 0; l < m; ++l) a_ += Ks[l * n + lane] * kf[m + l];
#pragma unroll
          for (int l = 0; l < m; ++l) a_ -= 2.0 * Ks[l * n + lane] * qv[n + l];
          red[64 + lane] = a_;
        }
      });
#pragma unroll
      for (int ks = 0; ks < KSM; ++ks) {
        double aop[Exec::SLOTS][MTN], bop[Exec::SLOTS][MTN];
        ex.each([&](int lane, int sl) {
          const int li = lane % 16, lk = 4 * ks + lane / 16;
          const int kc = (lk < m) ? lk : 0;
#pragma unroll
          for (int mt = 0; mt < MTN; ++mt) {
            const int i = 16 * mt + li;
            const double v = Q[((i < n) ? i : 0) * QLD + n + kc];
            aop[sl][mt] = (i < n && lk < m) ? -v : 0.0;
            const double w = Ks[kc * n + ((i < n) ? i : 0)];  // same index range for the K columns
            bop[sl][mt] = (i < n && lk < m) ? w : 0.0;
          }
        });
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < MTN; ++nt) ex.mfma(aop, mt, bop, nt, accQ, mt, nt);
      }
      ex.each([&](int lane, int sl) {
        const int lj = lane % 16, lq = lane / 16;
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < MTN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * mt + 4 * r + lq, j = 16 * nt + lj;
              smem[(i < n && j < n) ? SM::OFF_W + i * WLD + j : SM::OFF_DUMP] = accQ[sl][mt][nt][r];
            }
      });
      ex.sync();
      BWD_STAMP(4);
      // symmetrise + regularise -> V; NaN / overflow guard folded into the same pass
      const bool badV = ex.any([&](int lane, int sl) {
        bool bad = false;
        for (int i = lane; i < n * n; i += NL) {
          const int rr = i / n, cc = i % n;
          const double v_ = 0.5 * (W[rr * WLD + cc] + W[cc * WLD + rr]) + ((rr == cc) ? xreg : 0.0);
          V[i] = v_;
          bad = bad || bad_number(v_);
        }
        return bad;
      });
      BWD_STAMP(5);
      // gap contribution: Vx += Vxx f ; sums for the expected improvement
      ex.each([&](int lane, int sl) {
        if (lane >= n) return;
        double a_ = 0;
        if (infeas)
          for (int j = 0; j < n; ++j) a_ += V[lane * n + j] * rec[DM::OFF_GAP + j];
        const double nv = red[64 + lane] + (infeas ? a_ : 0.0);
        vx[lane] = nv;
        D.Vf[((size_t)b * (T + 1) + t) * n + lane] = a_;
        D.Vx[((size_t)b * (T + 1) + t) * n + lane] = nv;
        red[lane] = infeas ? nv * rec[DM::OFF_GAP + lane] : 0.0;
        red[32 + lane] = infeas ? rec[DM::OFF_GAP + lane] * a_ : 0.0;
      });
      ex.sync();
      double mxv = 0;
      bool nanv = false;
      for (int i = 0; i < n; ++i) {
        mxv = fmax(mxv, fabs(vx[i]));
        nanv = nanv || (vx[i] != vx[i]);
        dg_f -= red[i];
        dq_f += red[32 + i];
      }
      if (badV || nanv || bad_number(mxv)) {
        fail = true;
        break;
      }
      BWD_STAMP(6);
    }
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    if (b == 0)
      ex.each([&](int lane, int sl) {
        if (lane == 0)
          for (int i = 0; i < 8; ++i) D.dbg[16 + i] = bst[i];
      });
#endif
    ex.sync();
    if (!fail) break;
    xreg *= P.prm.reg_incfactor;
    if (xreg > P.prm.reg_max) xreg = P.prm.reg_max;
    ureg = xreg;
    if (xreg == P.prm.reg_max) {
      failed_final = true;
      break;
    }
  }
  ex.each([&](int lane, int sl) {
    if (lane == 0) {
      st.cost = cost;
      st.gapnorm = gapnorm;
      st.is_feasible = is_feasible;
      st.xreg = xreg;
      st.ureg = ureg;
      st.dg_u = dg_u;
      st.dq_u = dq_u;
      st.dg_f = dg_f;
      st.dq_f = dq_f;
      st.qu2 = qu2;
      st.bwd_failed = failed_final ? 1 : 0;
    }
  });
}

}  // namespace empc
