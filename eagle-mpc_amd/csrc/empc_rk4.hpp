// empc_rk4.hpp -- IntegratedActionModelRK4 inside the OCP (reference: src/factory/int-action.cpp:29-31, selected by the
// YAML keys `integrator` / `integration_method`; crocoddyl ~1.8 integ-action/rk4.hxx; oracle: oracle/action.hpp
// node_calc_rk4).  calcDiff of an RK4 node = four evaluations of the differential model + the chain rule through the
// stages.  On the device that is three steps per sweep:
//   1. rk4_stage_thread      one lane per (trajectory, knot): the stage states y_i = x (+) c_i dt k_{i-1} and the stage
//                            accelerations / contact forces, written as a "stage batch" of 4 B trajectories
//   2. the linearize kernel  unchanged code, run on the stage batch in RAW mode: differential-model derivatives
//                            da/dy, da/du and UNSCALED cost derivatives at (y_i, u), one record per stage
//   3. rk4_assemble_unit     one wavefront per (trajectory, knot): dy_i/dx, dk_i/dx, ... and the Gauss-Newton cost terms
//                            -> the node's record [Fx Fu | Lxx Lxu | Luu | Lx | Lu | gap | cost] in the tape the backward
//                            pass reads
// Rollouts of RK4 problems use node_nominal_rk4 (empc_dev_model.hpp) through the per-lane rollout kernel.
#pragma once
#include "empc_kernels.hpp"

namespace empc {

// buffers of the stage batch (index b' = 4 b + i)
struct Rk4Buffers {
  double* ys;     // [4B][T+1][NX]   stage states
  double* accs;   // [4B][T+1][NACC] stage accelerations | contact forces
  double* us4;    // [4B][T][NU]     the node's control, once per stage
  double* tape4;  // [4B][T+1][REC]  raw records
  TrajState* st4; // [4B]            copies of the trajectory state (the linearize kernel reads phase / smooth / flags)
};

// ---- step 1 -----------------------------------------------------------------------------------------------------------
template <class DM, int CT>
EMPC_HD void rk4_stage_thread(const DevBuffers& D, const Rk4Buffers& R, int b, int t) {
  constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NQ = DM::NQ, NDX = DM::NDX;
  const TrajState& st = D.st[b];
  const int T = D.T;
  if (t == 0)
    for (int i = 0; i < 4; ++i) R.st4[4 * b + i] = st;
  if (st.phase == PHASE_DONE || !st.need_lin) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
  const bool terminal = (t == T);
  const double* x = D.xs + ((size_t)b * (T + 1) + t) * NX;
  const double* u = terminal ? nullptr : D.us + ((size_t)b * T + t) * NU;
  const double rk4_c[4] = {0.0, 0.5, 0.5, 1.0};
  const double dt = P.dt;
  double y[NX], kprev[NDX];
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = x[i];
  for (int sg = 0; sg < 4; ++sg) {
    if (sg > 0) {
      double dxr[NDX];
#pragma unroll
      for (int j = 0; j < NDX; ++j) dxr[j] = rk4_c[sg] * dt * kprev[j];
      state_integrate<DM>(x, dxr, y, nullptr);
    }
    double a_[NV], us_[NU], lam_[6], ell_;
    dam_nominal<DM, CT>(P, set, st.smooth, y, u, terminal, (double*)nullptr, a_, ell_, us_, lam_);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      kprev[j] = y[NQ + j];
      kprev[NV + j] = a_[j];
    }
    const size_t bs = (size_t)4 * b + sg;
    double* yo = R.ys + (bs * (T + 1) + t) * NX;
    double* ao = R.accs + (bs * (T + 1) + t) * DM::NACC;
#pragma unroll
    for (int i = 0; i < NX; ++i) yo[i] = y[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) ao[i] = a_[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) ao[NV + i] = lam_[i];
    if (!terminal) {
      double* uo = R.us4 + (bs * T + t) * NU;
#pragma unroll
      for (int i = 0; i < NU; ++i) uo[i] = u[i];
    }
  }
}

// ---- step 3 -----------------------------------------------------------------------------------------------------------
template <class DM>
struct Rk4Smem {
  static constexpr int n = DM::NDX, m = DM::NU, nv = DM::NV;
  static constexpr int OFF_J = 0;                      // 4 x (J1 36 | J2 36): stages 1..3 and the final step
  static constexpr int OFF_DYX = OFF_J + 4 * 72;       // dy_i/dx      n x n
  static constexpr int OFF_DYU = OFF_DYX + n * n;      // dy_i/du      n x m
  static constexpr int OFF_DKX = OFF_DYU + n * m;      // dk_{i}/dx    n x n
  static constexpr int OFF_DKU = OFF_DKX + n * n;      // dk_{i}/du    n x m
  static constexpr int OFF_XX = OFF_DKU + n * m;       // lxx_i dy_i/dx
  static constexpr int OFF_XU = OFF_XX + n * n;        // lxx_i dy_i/du
  static constexpr int OFF_RAW = OFF_XU + n * m;       // the stage's raw record
  static constexpr int OFF_SKX = OFF_RAW + (DM::REC + 63) / 64 * 64;  // sum w_i dk_i/dx, then Fx
  static constexpr int OFF_SKU = OFF_SKX + n * n;
  static constexpr int OFF_LXX = OFF_SKU + n * m;      // accumulators of the node's cost derivatives
  static constexpr int OFF_LXU = OFF_LXX + n * n;
  static constexpr int OFF_LUU = OFF_LXU + n * m;
  static constexpr int OFF_LX = OFF_LUU + m * m;
  static constexpr int OFF_LU = OFF_LX + n;
  static constexpr int OFF_K = OFF_LU + m;             // k_0..k_3 (4 x n), dx (n), cost parts (4)
  static constexpr int SIZE = (OFF_K + 5 * n + 4 + 1) / 2 * 2;
};

// blkdiag(J (6 x 6), I) from the left on an n x cols matrix held in LDS, in place; all lanes cooperate (one column each)
template <class Exec>
EMPC_HD void rk4_apply_block(Exec& ex, const double* J6, double* M, int cols, int nl) {
  ex.each([&](int lane, int sl) {
    for (int j = lane; j < cols; j += nl) {
      double col[6], out[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) col[i] = M[i * cols + j];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double a_ = 0;
#pragma unroll
        for (int l = 0; l < 6; ++l) a_ += J6[i * 6 + l] * col[l];
        out[i] = a_;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) M[i * cols + j] = out[i];
    }
  });
  ex.sync();
}

// Jintegrate blocks of StateMultibody at step d: J1 = Ad(exp6(d)^-1) (derivative w.r.t. x), J2 = Jexp6(d) (w.r.t. d)
EMPC_HD void rk4_jint_blocks(const double* d, double* J1, double* J2) {
  double qe[4], pe[3], Re[9], Px[9], RtP[9];
  exp6_quat(d, qe, pe);
  Jexp6(d, pe, J2);
  quat_to_R(qe, Re);
  skew3(pe, Px);
  matTmul3<double>(Re, Px, RtP);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      J1[r * 6 + c] = Re[3 * c + r];
      J1[r * 6 + 3 + c] = -RtP[3 * r + c];
      J1[(3 + r) * 6 + c] = 0.0;
      J1[(3 + r) * 6 + 3 + c] = Re[3 * c + r];
    }
}

// One (trajectory, knot) unit, nl lanes (one wavefront).  Raw records: da/dy in rows nv.. of the Fx block, da/du in rows
// nv.. of the Fu block, cost blocks unscaled (linearize in RAW mode).
template <class DM, class Exec>
EMPC_HD void rk4_assemble_unit(Exec& ex, const DevBuffers& D, const Rk4Buffers& R, int b, int t, int nl, double* N) {
  typedef Rk4Smem<DM> SM;
  constexpr int n = DM::NDX, m = DM::NU, nv = DM::NV, NX = DM::NX, NQ = DM::NQ, REC = DM::REC, NM = DM::NM;
  const TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE || !st.need_lin) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const int T = D.T;
  const bool terminal = (t == T);
  const bool feas = st.is_feasible != 0;
  const double dt = P.dt;
  const double rk4_c[4] = {0.0, 0.5, 0.5, 1.0};
  double* Jb = N + SM::OFF_J;
  double* DYX = N + SM::OFF_DYX;
  double* DYU = N + SM::OFF_DYU;
  double* DKX = N + SM::OFF_DKX;
  double* DKU = N + SM::OFF_DKU;
  double* XX = N + SM::OFF_XX;
  double* XU = N + SM::OFF_XU;
  double* RAW = N + SM::OFF_RAW;
  double* SKX = N + SM::OFF_SKX;
  double* SKU = N + SM::OFF_SKU;
  double* LXX = N + SM::OFF_LXX;
  double* LXU = N + SM::OFF_LXU;
  double* LUU = N + SM::OFF_LUU;
  double* LX = N + SM::OFF_LX;
  double* LU = N + SM::OFF_LU;
  double* KK = N + SM::OFF_K;
  const double* xg = D.xs + ((size_t)b * (T + 1) + t) * NX;
  double* out = D.tape + ((size_t)b * (T + 1) + t) * REC;

  // k_i = [v(y_i); a_i], dx = dt/6 (k0 + 2 k1 + 2 k2 + k3); zero the accumulators
  ex.each([&](int lane, int sl) {
    for (int e = lane; e < 4 * n; e += nl) {
      const int i = e / n, j = e % n;
      const size_t bs = (size_t)4 * b + i;
      KK[e] = (j < nv) ? R.ys[(bs * (T + 1) + t) * NX + NQ + j] : R.accs[(bs * (T + 1) + t) * DM::NACC + (j - nv)];
    }
    for (int e = lane; e < n * n; e += nl) {
      SKX[e] = 0.0;
      LXX[e] = 0.0;
    }
    for (int e = lane; e < n * m; e += nl) {
      SKU[e] = 0.0;
      LXU[e] = 0.0;
    }
    for (int e = lane; e < m * m; e += nl) LUU[e] = 0.0;
    if (lane < n) LX[lane] = 0.0;
    if (lane < m) LU[lane] = 0.0;
  });
  ex.sync();
  ex.each([&](int lane, int sl) {
    if (lane < n) KK[4 * n + lane] = (KK[lane] + 2.0 * KK[n + lane] + 2.0 * KK[2 * n + lane] + KK[3 * n + lane]) * dt / 6.0;
  });
  ex.sync();
  // the four pairs of Jintegrate blocks (stages 1..3 at c_i dt k_{i-1}, final step at dx), one lane each
  ex.each([&](int lane, int sl) {
    if (lane < 4) {
      double d[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) d[j] = (lane < 3) ? rk4_c[lane + 1] * dt * KK[lane * n + j] : KK[4 * n + j];
      rk4_jint_blocks(d, Jb + lane * 72, Jb + lane * 72 + 36);
    }
  });
  ex.sync();

  for (int sg = 0; sg < 4; ++sg) {
    const double w = (sg == 0 || sg == 3) ? 1.0 : 2.0;
    // raw record of the stage -> LDS
    {
      const double* rr = R.tape4 + (((size_t)4 * b + sg) * (T + 1) + t) * REC;
      ex.each([&](int lane, int sl) {
        for (int e = lane; e < REC; e += nl) RAW[e] = rr[e];
      });
    }
    // dy_i/dx, dy_i/du
    if (sg == 0) {
      ex.each([&](int lane, int sl) {
        for (int e = lane; e < n * n; e += nl) DYX[e] = ((e / n) == (e % n)) ? 1.0 : 0.0;
        for (int e = lane; e < n * m; e += nl) DYU[e] = 0.0;
      });
      ex.sync();
    } else {
      const double cdt = rk4_c[sg] * dt;
      ex.each([&](int lane, int sl) {
        for (int e = lane; e < n * n; e += nl) DYX[e] = cdt * DKX[e];
        for (int e = lane; e < n * m; e += nl) DYU[e] = cdt * DKU[e];
      });
      ex.sync();
      const double* J1 = Jb + (sg - 1) * 72;
      const double* J2 = J1 + 36;
      rk4_apply_block(ex, J2, DYX, n, nl);
      rk4_apply_block(ex, J2, DYU, m, nl);
      ex.each([&](int lane, int sl) {
        for (int e = lane; e < n * n; e += nl) {
          const int r = e / n, q = e % n;
          DYX[e] += (r < 6 && q < 6) ? J1[r * 6 + q] : ((r >= 6 && r == q) ? 1.0 : 0.0);
        }
      });
      ex.sync();
    }
    // dk_i/dx, dk_i/du: velocity rows are rows of dy_i, acceleration rows A_i dy_i (+ B_i)
    ex.each([&](int lane, int sl) {
      for (int e = lane; e < n * n; e += nl) {
        const int r = e / n, q = e % n;
        double v_;
        if (r < nv) {
          v_ = DYX[(nv + r) * n + q];
        } else {
          v_ = 0;
          for (int l = 0; l < n; ++l) v_ += RAW[DM::OFF_FX + r * NM + l] * DYX[l * n + q];
        }
        DKX[e] = v_;
      }
      for (int e = lane; e < n * m; e += nl) {
        const int r = e / m, q = e % m;
        double v_;
        if (r < nv) {
          v_ = DYU[(nv + r) * m + q];
        } else {
          v_ = RAW[DM::OFF_FU + r * NM + q];
          for (int l = 0; l < n; ++l) v_ += RAW[DM::OFF_FX + r * NM + l] * DYU[l * m + q];
        }
        DKU[e] = v_;
      }
      // lxx_i dy_i/dx, lxx_i dy_i/du
      for (int e = lane; e < n * n; e += nl) {
        const int r = e / n, q = e % n;
        double v_ = 0;
        for (int l = 0; l < n; ++l) v_ += RAW[DM::OFF_LXX + r * NM + l] * DYX[l * n + q];
        XX[e] = v_;
      }
      for (int e = lane; e < n * m; e += nl) {
        const int r = e / m, q = e % m;
        double v_ = 0;
        for (int l = 0; l < n; ++l) v_ += RAW[DM::OFF_LXX + r * NM + l] * DYU[l * m + q];
        XU[e] = v_;
      }
    });
    ex.sync();
    // accumulate
    ex.each([&](int lane, int sl) {
      for (int e = lane; e < n * n; e += nl) {
        const int r = e / n, q = e % n;
        SKX[e] += w * DKX[e];
        double v_ = 0;
        for (int l = 0; l < n; ++l) v_ += DYX[l * n + r] * XX[l * n + q];
        LXX[e] += w * v_;
      }
      for (int e = lane; e < n * m; e += nl) {
        const int r = e / m, q = e % m;
        SKU[e] += w * DKU[e];
        double v_ = 0;
        for (int l = 0; l < n; ++l) v_ += DYX[l * n + r] * (RAW[DM::OFF_LXU + l * NM + q] + XU[l * m + q]);
        LXU[e] += w * v_;
      }
      for (int e = lane; e < m * m; e += nl) {
        const int r = e / m, q = e % m;
        double v_ = RAW[DM::OFF_LUU + r * m + q];
        for (int l = 0; l < n; ++l)
          v_ += RAW[DM::OFF_LXU + l * NM + r] * DYU[l * m + q] + DYU[l * m + r] * RAW[DM::OFF_LXU + l * NM + q] + DYU[l * m + r] * XU[l * m + q];
        LUU[e] += w * v_;
      }
      if (lane < n) {
        double v_ = 0;
        for (int l = 0; l < n; ++l) v_ += DYX[l * n + lane] * RAW[DM::OFF_LX + l];
        LX[lane] += w * v_;
      } else if (lane - n < m && lane >= n) {
        const int q = lane - n;
        double v_ = RAW[DM::OFF_LU + q];
        for (int l = 0; l < n; ++l) v_ += DYU[l * m + q] * RAW[DM::OFF_LX + l];
        LU[q] += w * v_;
      }
      if (lane == nl - 1) KK[5 * n + sg] = RAW[DM::OFF_COST];
    });
    ex.sync();
  }
  // Fx = blk(J2) (dt/6 sum w dk/dx) + blk(J1, I), Fu = blk(J2) (dt/6 sum w dk/du)
  ex.each([&](int lane, int sl) {
    for (int e = lane; e < n * n; e += nl) SKX[e] = SKX[e] * dt / 6.0;
    for (int e = lane; e < n * m; e += nl) SKU[e] = SKU[e] * dt / 6.0;
  });
  ex.sync();
  {
    const double* J1 = Jb + 3 * 72;
    const double* J2 = J1 + 36;
    rk4_apply_block(ex, J2, SKX, n, nl);
    rk4_apply_block(ex, J2, SKU, m, nl);
    const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 / 6.0 : dt / 6.0;
    ex.each([&](int lane, int sl) {
      for (int e = lane; e < n * n; e += nl) {
        const int r = e / n, q = e % n;
        out[DM::OFF_FX + r * NM + q] = SKX[e] + ((r < 6 && q < 6) ? J1[r * 6 + q] : ((r >= 6 && r == q) ? 1.0 : 0.0));
        out[DM::OFF_LXX + r * NM + q] = LXX[e] * cscale;
      }
      for (int e = lane; e < n * m; e += nl) {
        const int r = e / m, q = e % m;
        out[DM::OFF_FU + r * NM + q] = SKU[e];
        out[DM::OFF_LXU + r * NM + q] = LXU[e] * cscale;
      }
      for (int e = lane; e < m * m; e += nl) out[DM::OFF_LUU + e] = LUU[e] * cscale;
      if (lane < n) out[DM::OFF_LX + lane] = LX[lane] * cscale;
      if (lane < m) out[DM::OFF_LU + lane] = LU[lane] * cscale;
      if (lane == 0) out[DM::OFF_COST] = (KK[5 * n] + 2.0 * KK[5 * n + 1] + 2.0 * KK[5 * n + 2] + KK[5 * n + 3]) * cscale;
      // gaps: fs[t+1] = xnext (-) xs[t+1];  fs[0] = x0 (-) xs[0]
      if (lane == 1 && !terminal) {
        double gap[n];
        if (!feas) {
          double x[NX], xn[NX], dxv[n];
#pragma unroll
          for (int i = 0; i < NX; ++i) x[i] = xg[i];
#pragma unroll
          for (int i = 0; i < n; ++i) dxv[i] = KK[4 * n + i];
          state_integrate<DM>(x, dxv, xn, nullptr);
          state_diff<DM>(xg + NX, xn, gap, nullptr);
        }
        double* go = D.tape + ((size_t)b * (T + 1) + t + 1) * REC + DM::OFF_GAP;
#pragma unroll
        for (int i = 0; i < n; ++i) go[i] = feas ? 0.0 : gap[i];
      }
      if (lane == 2 && t == 0) {
        double gap[n];
        if (!feas) state_diff<DM>(xg, D.x0 + (size_t)b * NX, gap, nullptr);
        double* go = D.tape + ((size_t)b * (T + 1)) * REC + DM::OFF_GAP;
#pragma unroll
        for (int i = 0; i < n; ++i) go[i] = feas ? 0.0 : gap[i];
      }
    });
  }
}

}  // namespace empc
