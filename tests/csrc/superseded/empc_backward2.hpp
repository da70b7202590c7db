// empc_backward2.hpp -- HOT-B kernel body, second generation: one WORKGROUP (NL = 64 * NW lanes) per trajectory.
//
// The Riccati sweep is sequential in t; per node the work is a handful of small dense products.  Four wavefronts
// share them (column x row-group tiling, operands staged in LDS, each lane keeping its operand column in registers),
// the next node's tape record is prefetched into registers while the current one is processed, and the serial part
// (LLT of Quu and the gain solves) runs in wavefront 0 while the other wavefronts of the CU's other workgroups fill
// the SIMDs.  Semantics are those of backward_traj (v1) and of the oracle: SolverDDP::backwardPass + computeGains +
// SolverFDDP::updateExpectedImprovement, with the regularisation retry loop of src/sbfddp.cpp:242-255.
#pragma once
// TEST INFRASTRUCTURE (superseded kernel form, kept as a cross-check of the shipped one through the CPU lane emulator;
// not compiled into libempc.so)
#include "../../../eagle-mpc_amd/csrc/empc_kernels.hpp"

namespace empc {

template <class DM>
struct Bwd2Smem {
  static constexpr int n = DM::NDX, m = DM::NU, nm = n + m;
  static constexpr int OFF_REC = 0;
  static constexpr int OFF_V = DM::REC;
  static constexpr int OFF_VX = OFF_V + n * n;
  static constexpr int OFF_W = OFF_VX + n;        // n x nm
  static constexpr int OFF_Q = OFF_W + n * nm;    // nm x nm
  static constexpr int OFF_QV = OFF_Q + nm * nm;  // nm
  static constexpr int OFF_K = OFF_QV + nm;       // m x n
  static constexpr int OFF_KF = OFF_K + m * n;    // k (m), Quuk (m)
  static constexpr int OFF_RED = OFF_KF + 2 * m;  // 4 x 32 partial sums
  static constexpr int OFF_FLAG = OFF_RED + 128;
  static constexpr int OFF_PRO = OFF_FLAG + 8;    // prologue reductions: 3 x 256
  static constexpr int SIZE = (OFF_PRO + 3 * 256 + 1) / 2 * 2;
};

// Exec concept additions used here: ex.sync() is a barrier over all NL lanes, ex.any(f) an OR-reduction + barrier.
template <class DM, int NL, class Exec>
EMPC_HD void backward_traj2(Exec& ex, const DevBuffers& D, int b, double* smem) {
  typedef Bwd2Smem<DM> SM;
  constexpr int n = DM::NDX, m = DM::NU, nm = n + m, REC = DM::REC;
  constexpr int CW = 32;                    // column groups of 32 lanes (nm <= 32) ...
  static_assert(nm <= 64, "state + control dimension too large for the column tiling");
  constexpr int CWE = (nm <= 32) ? 32 : 64; // ... or 64 lanes for the largest robots
  constexpr int NG = NL / CWE;              // row groups
  constexpr int PRE = (REC + NL - 1) / NL;  // prefetch registers per lane
#ifndef EMPC_BWD_RP
#define EMPC_BWD_RP 2
#endif
  constexpr int RP = EMPC_BWD_RP;           // rows per pass of the product stages
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
      for (int q = 0; q < PRE; ++q) {
        const int i = lane + q * NL;
        pre[sl][q] = (i < REC) ? r[i] : 0.0;
      }
    });
    for (int t = T - 1; t >= 0; --t) {
      ex.each([&](int lane, int sl) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
          const int i = lane + q * NL;
          if (i < REC) rec[i] = pre[sl][q];
        }
        if (t > 0) {
          const double* r = tape + (size_t)(t - 1) * REC;
#pragma unroll
          for (int q = 0; q < PRE; ++q) {
            const int i = lane + q * NL;
            pre[sl][q] = (i < REC) ? r[i] : 0.0;
          }
        }
      });
      ex.sync();
      // W = V' A, A = [Fx Fu]
      ex.each([&](int lane, int sl) {
        const int c = lane % CWE, g = lane / CWE;
        if (c >= nm) return;
        double Acol[n];
#pragma unroll
        for (int k2 = 0; k2 < n; ++k2) Acol[k2] = rec[DM::OFF_A + k2 * nm + c];
        // RP rows per pass: RP independent accumulation chains hide each other's LDS and FMA latency
        for (int i = g; i < n; i += RP * NG) {
          int ri[RP];
          double acc[RP];
#pragma unroll
          for (int r = 0; r < RP; ++r) {
            ri[r] = (i + r * NG < n) ? i + r * NG : i;  // surplus rows recompute row i and are not stored
            acc[r] = 0.0;
          }
#pragma unroll
          for (int k2 = 0; k2 < n; ++k2) {
#pragma unroll
            for (int r = 0; r < RP; ++r) acc[r] += V[ri[r] * n + k2] * Acol[k2];
          }
          W[i * nm + c] = acc[0];
#pragma unroll
          for (int r = 1; r < RP; ++r)
            if (ri[r] != i) W[ri[r] * nm + c] = acc[r];
        }
      });
      ex.sync();
      // Q = H + A^T W (blocks xx, xu, uu), qv = [Lx; Lu] + A^T vx'
      ex.each([&](int lane, int sl) {
        const int c = lane % CWE, g = lane / CWE;
        if (c >= nm) return;
        double Wcol[n];
#pragma unroll
        for (int k2 = 0; k2 < n; ++k2) Wcol[k2] = W[k2 * nm + c];
        // RP rows per pass (see W); rows of the uu block are skipped for the x columns, as before
        const int rmax = (c < n) ? n : nm;
        for (int rr = g; rr < rmax; rr += RP * NG) {
          int ri[RP];
          double acc[RP];
#pragma unroll
          for (int r = 0; r < RP; ++r) {
            ri[r] = (rr + r * NG < rmax) ? rr + r * NG : rr;
            acc[r] = (ri[r] < n) ? rec[DM::OFF_HX + ri[r] * nm + c] : rec[DM::OFF_LUU + (ri[r] - n) * m + (c - n)];
          }
#pragma unroll
          for (int k2 = 0; k2 < n; ++k2) {
#pragma unroll
            for (int r = 0; r < RP; ++r) acc[r] += rec[DM::OFF_A + k2 * nm + ri[r]] * Wcol[k2];
          }
          Q[rr * nm + c] = acc[0];
#pragma unroll
          for (int r = 1; r < RP; ++r)
            if (ri[r] != rr) Q[ri[r] * nm + c] = acc[r];
        }
        if (g == NG - 1) {
          double a_ = (c < n) ? rec[DM::OFF_LX + c] : rec[DM::OFF_LU + (c - n)];
#pragma unroll
          for (int k2 = 0; k2 < n; ++k2) {
            a_ += rec[DM::OFF_A + k2 * nm + c] * vx[k2];
          }
          qv[c] = a_;
        }
      });
      ex.sync();
      // computeGains in the first wavefront: LLT(Quu + ureg I); K = Quu^-1 Qxu^T ; k = Quu^-1 Qu ; Quuk
      ex.each([&](int lane, int sl) {
        if (lane >= 64) return;
        double Lq[m * (m + 1) / 2];
#pragma unroll
        for (int i = 0; i < m; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) Lq[i * (i + 1) / 2 + j] = Q[(n + i) * nm + n + j] + ((i == j) ? ureg : 0.0);
        const bool pd = chol_packed<m>(Lq);
        if (lane == 0) flag[0] = pd ? 0.0 : 1.0;
        if (lane <= n) {
          double rhs[m];
#pragma unroll
          for (int i = 0; i < m; ++i) rhs[i] = (lane < n) ? Q[lane * nm + n + i] : qv[n + i];
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
              for (int j = 0; j < m; ++j) a_ += Q[(n + (i > j ? i : j)) * nm + n + (i > j ? j : i)] * rhs[j];
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
      // gains out; Vxx = Qxx - Qxu K (into W), Vx = Qx + K^T Quuk - 2 K^T Qu (into red)
      ex.each([&](int lane, int sl) {
        double* Kg = D.K + ((size_t)b * T + t) * m * n;
        for (int i = lane; i < m * n; i += NL) Kg[i] = Ks[i];
        if (lane < m) D.kff[((size_t)b * T + t) * m + lane] = kf[lane];
        const int c = lane % CWE, g = lane / CWE;
        if (c < n) {
          double Kcol[m];
#pragma unroll
          for (int l = 0; l < m; ++l) Kcol[l] = Ks[l * n + c];
          for (int i = g; i < n; i += RP * NG) {
            int ri[RP];
            double acc[RP];
#pragma unroll
            for (int r = 0; r < RP; ++r) {
              ri[r] = (i + r * NG < n) ? i + r * NG : i;
              acc[r] = Q[ri[r] * nm + c];
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {
#pragma unroll
              for (int r = 0; r < RP; ++r) acc[r] -= Q[ri[r] * nm + n + l] * Kcol[l];
            }
            W[i * nm + c] = acc[0];
#pragma unroll
            for (int r = 1; r < RP; ++r)
              if (ri[r] != i) W[ri[r] * nm + c] = acc[r];
          }
          if (g == NG - 1) {
            double a_ = qv[c];
#pragma unroll
            for (int l = 0; l < m; ++l) a_ += Kcol[l] * kf[m + l];
#pragma unroll
            for (int l = 0; l < m; ++l) a_ -= 2.0 * Kcol[l] * qv[n + l];
            red[64 + c] = a_;
          }
        }
      });
      ex.sync();
      // symmetrise + regularise -> V; NaN / overflow guard folded into the same pass
      const bool badV = ex.any([&](int lane, int sl) {
        bool bad = false;
        for (int i = lane; i < n * n; i += NL) {
          const int rr = i / n, cc = i % n;
          const double v_ = 0.5 * (W[rr * nm + cc] + W[cc * nm + rr]) + ((rr == cc) ? xreg : 0.0);
          V[i] = v_;
          bad = bad || bad_number(v_);
        }
        return bad;
      });
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
    }
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
