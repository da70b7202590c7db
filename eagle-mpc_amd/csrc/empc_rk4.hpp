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
// The chain rule through the four stages is dense n x n / n x (n + m) algebra per node: every product runs on the matrix
// cores (v_mfma_f64_16x16x4_f64, the operand conventions of the backward pass), its operands in zero-padded LDS arrays with
// an odd row stride.  Per stage i, with DY = [dy_i/dx | dy_i/du] (n x nm):
//   DK rows nv..   = A_i DY (+ B_i on the control columns)          A_i = da/dy, B_i = da/du of the stage's raw record
//   Z              = w_i (lxx_i DY + [0 | lxu_i])
//   [Lxx Lxu; . Luu] += DY^T Z,   Luu += w_i (luu_i + lxu_i^T DYU)  (the transposed term that DY^T Z lacks)
// Before: every element of every product one dot product out of LDS (two LDS reads per multiply-add, 39 KB of LDS per
// wavefront = one wavefront per SIMD): 7.7 ms per launch at B = 1024, the longest kernel of an RK4 sweep.
template <class DM>
struct Rk4Smem {
  static constexpr int n = DM::NDX, m = DM::NU, nv = DM::NV, nm = n + m;
  static constexpr int NT = (nm + 15) / 16;   // column tiles of [x | u]
  static constexpr int MTX = (n + 15) / 16;   // row tiles of an n-row product
  static constexpr int MTA = (nv + 15) / 16;  // row tiles of the acceleration rows
  static constexpr int MTU = (m + 15) / 16;
  static constexpr int KS = (n + 3) / 4;      // k steps over the state dimension
  static constexpr int KR = 4 * KS;           // rows of DY / Z / DK (rows >= n stay zero)
  static constexpr int LD = 16 * NT + 1;      // their row stride (columns >= nm stay zero)
  static constexpr int PLD = 16 * MTU + 1;
  static constexpr int OFF_J = 0;                       // 4 x (J1 36 | J2 36): stages 1..3 and the final step
  static constexpr int OFF_DY = OFF_J + 4 * 72;
  static constexpr int OFF_Z = OFF_DY + KR * LD;
  static constexpr int OFF_DK = OFF_Z + KR * LD;
  static constexpr int OFF_P = OFF_DK + KR * LD;        // the Luu tile that is not part of DY^T Z, for the store
  static constexpr int OFF_K = OFF_P + 16 * MTU * PLD;  // k_0..k_3 (4 x n), dx (n), cost parts (4)
  static constexpr int OFF_RAW = (OFF_K + 5 * n + 4 + 1) / 2 * 2;  // the stage's raw record
  static constexpr int SIZE = (OFF_RAW + (DM::REC + 63) / 64 * 64 + 1) / 2 * 2;
  static constexpr int NSK = (n * nm + 63) / 64;        // elements of sum w_i dk_i per lane
};

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

// One (trajectory, knot) unit on one wavefront (nl = 64 lanes).  Raw records: da/dy in rows nv.. of the Fx block, da/du in rows
// nv.. of the Fu block, cost blocks unscaled (linearize in RAW mode).
template <class DM, class Exec>
EMPC_HD void rk4_assemble_unit(Exec& ex, const DevBuffers& D, const Rk4Buffers& R, int b, int t, int nl, double* N) {
  typedef Rk4Smem<DM> SM;
  constexpr int n = DM::NDX, m = DM::NU, nv = DM::NV, nm = n + m, NX = DM::NX, NQ = DM::NQ, REC = DM::REC, NM = DM::NM;
  constexpr int NT = SM::NT, MTX = SM::MTX, MTA = SM::MTA, MTU = SM::MTU, KS = SM::KS, LD = SM::LD, PLD = SM::PLD;
  static_assert(NM == nm, "the record's row stride is n + m");
  const TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE || !st.need_lin) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const int T = D.T;
  const bool terminal = (t == T);
  const bool feas = st.is_feasible != 0;
  const double dt = P.dt;
  const double rk4_c[4] = {0.0, 0.5, 0.5, 1.0};
  double* Jb = N + SM::OFF_J;
  double* DY = N + SM::OFF_DY;
  double* Z = N + SM::OFF_Z;
  double* DK = N + SM::OFF_DK;
  double* PT = N + SM::OFF_P;
  double* KK = N + SM::OFF_K;
  double* RAW = N + SM::OFF_RAW;
  const double* xg = D.xs + ((size_t)b * (T + 1) + t) * NX;
  double* out = D.tape + ((size_t)b * (T + 1) + t) * REC;

  // accumulators that live across the stages: the cost Hessian tiles, the extra Luu tile, Lx | Lu (lane = column), sum w dk
  double accL[Exec::SLOTS][NT][NT][4], accP[Exec::SLOTS][MTU][MTU][4], lxu[Exec::SLOTS], sk[Exec::SLOTS][SM::NSK];
  ex.each([&](int lane, int sl) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) accL[sl][a][c][r] = 0.0;
#pragma unroll
    for (int a = 0; a < MTU; ++a)
#pragma unroll
      for (int c = 0; c < MTU; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) accP[sl][a][c][r] = 0.0;
    lxu[sl] = 0.0;
#pragma unroll
    for (int q = 0; q < SM::NSK; ++q) sk[sl][q] = 0.0;
    // k_i = [v(y_i); a_i]; the padded operand arrays start from zero (rows >= n and columns >= nm are never written)
    for (int e = lane; e < 4 * n; e += nl) {
      const int i = e / n, j = e % n;
      const size_t bs = (size_t)4 * b + i;
      KK[e] = (j < nv) ? R.ys[(bs * (T + 1) + t) * NX + NQ + j] : R.accs[(bs * (T + 1) + t) * DM::NACC + (j - nv)];
    }
    for (int e = lane; e < 3 * SM::KR * LD; e += nl) DY[e] = 0.0;  // DY, Z, DK are contiguous
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

  // the raw record of the next stage travels in registers while the current stage is worked on (one memory latency per
  // stage would otherwise sit in front of every stage of this one-wavefront-per-SIMD kernel)
  constexpr int NPRE = (REC + 63) / 64;
  double pre[Exec::SLOTS][NPRE];
  auto fetch_raw = [&](int sg, int lane, int sl) {
    const double* rr = R.tape4 + (((size_t)4 * b + sg) * (T + 1) + t) * REC;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int e = lane + q * nl;
      pre[sl][q] = rr[e < REC ? e : REC - 1];
    }
  };
  ex.each([&](int lane, int sl) { fetch_raw(0, lane, sl); });
  for (int sg = 0; sg < 4; ++sg) {
    const double w = (sg == 0 || sg == 3) ? 1.0 : 2.0;
    // raw record of the stage -> LDS;  DY of the stage, one column per lane: blkdiag(J2, I) (c dt DK) + blkdiag(J1, I) | 0
    {
      const double cdt = rk4_c[sg] * dt;
      const double* J1 = Jb + (sg > 0 ? sg - 1 : 0) * 72;
      const double* J2 = J1 + 36;
      ex.each([&](int lane, int sl) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
          const int e = lane + q * nl;
          if (e < REC) RAW[e] = pre[sl][q];
        }
        if (sg < 3) fetch_raw(sg + 1, lane, sl);
        if (lane >= nm) return;
        const int c = lane;
        if (sg == 0) {
          for (int i = 0; i < n; ++i) DY[i * LD + c] = (i == c) ? 1.0 : 0.0;
        } else {
          double col[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) col[i] = cdt * DK[i * LD + c];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double a_ = 0;
#pragma unroll
            for (int l = 0; l < 6; ++l) a_ += J2[i * 6 + l] * col[l];
            DY[i * LD + c] = a_ + ((c < 6) ? J1[i * 6 + c] : 0.0);
          }
          for (int i = 6; i < n; ++i) DY[i * LD + c] = cdt * DK[i * LD + c] + ((i == c) ? 1.0 : 0.0);
        }
      });
    }
    ex.sync();
    // A_i DY and lxx_i DY
    double accA[Exec::SLOTS][MTA][NT][4], accB[Exec::SLOTS][MTX][NT][4];
    ex.each([&](int lane, int sl) {
#pragma unroll
      for (int a = 0; a < MTA; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) accA[sl][a][c][r] = 0.0;
#pragma unroll
      for (int a = 0; a < MTX; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) accB[sl][a][c][r] = 0.0;
    });
    {
      // (columns 4 ks + lq >= n of the A operands meet the zero rows of DY; rows beyond the block are clamped: their products
      //  land in rows nobody stores.  Operands of k step ks + 1 are requested from LDS before the matrix-core instructions of
      //  step ks are issued: the stage fences of `each` would otherwise expose one LDS round trip per k step)
      double aA[2][Exec::SLOTS][MTA], aB[2][Exec::SLOTS][MTX], bY[2][Exec::SLOTS][NT];
      auto load1 = [&](int ks, int buf) {
        ex.each([&](int lane, int sl) {
          const int li = lane % 16, lq = lane / 16;
#pragma unroll
          for (int a = 0; a < MTA; ++a) {
            const int row = nv + 16 * a + li;
            aA[buf][sl][a] = RAW[DM::OFF_FX + (row < n ? row : n - 1) * NM + 4 * ks + lq];
          }
#pragma unroll
          for (int a = 0; a < MTX; ++a) {
            const int row = 16 * a + li;
#if EMPC_REC_TRI
            aB[buf][sl][a] = RAW[DM::lxx(row < n ? row : n - 1, (4 * ks + lq) < n ? 4 * ks + lq : n - 1)];  // (columns >= n: any finite value)
#else
            aB[buf][sl][a] = RAW[DM::OFF_LXX + (row < n ? row : n - 1) * NM + 4 * ks + lq];
#endif
          }
#pragma unroll
          for (int c = 0; c < NT; ++c) bY[buf][sl][c] = DY[(4 * ks + lq) * LD + 16 * c + li];
        });
      };
      load1(0, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) load1(ks + 1, (ks + 1) & 1);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
          for (int a = 0; a < MTA; ++a) ex.mfma(aA[ks & 1], a, bY[ks & 1], c, accA, a, c);
#pragma unroll
          for (int a = 0; a < MTX; ++a) ex.mfma(aB[ks & 1], a, bY[ks & 1], c, accB, a, c);
        }
      }
    }
    // DK = [velocity rows of DY; A_i DY + B_i],  Z = w (lxx_i DY + [0 | lxu_i]),  Lx | Lu,  the raw Luu
    ex.each([&](int lane, int sl) {
      const int lj = lane % 16, lq = lane / 16;
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const int col = 16 * c + lj;
        if (col >= nm) continue;
#pragma unroll
        for (int a = 0; a < MTA; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * a + 4 * r + lq;
            if (row < nv)
              DK[(nv + row) * LD + col] = accA[sl][a][c][r] + ((col >= n) ? RAW[DM::OFF_FU + (nv + row) * NM + col - n] : 0.0);
          }
#pragma unroll
        for (int a = 0; a < MTX; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * a + 4 * r + lq;
            if (row < n) Z[row * LD + col] = w * (accB[sl][a][c][r] + ((col >= n) ? RAW[EMPC_REC_TRI ? DM::lxu(row, col - n) : DM::FULL_LXU + row * NM + col - n] : 0.0));
          }
      }
      for (int e = lane; e < nv * nm; e += nl) {
        const int r = e / nm, c = e % nm;
        DK[r * LD + c] = DY[(nv + r) * LD + c];
      }
#pragma unroll
      for (int a = 0; a < MTU; ++a)
#pragma unroll
        for (int c = 0; c < MTU; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * a + 4 * r + lq, j = 16 * c + lj;
            if (i < m && j < m) accP[sl][a][c][r] += w * RAW[EMPC_REC_TRI ? DM::luu(i, j) : DM::FULL_LUU + i * m + j];
          }
      if (lane < nm) {
        double v_ = (lane >= n) ? RAW[DM::OFF_LU + lane - n] : 0.0;
        double dy_[n], lx_[n];
#pragma unroll
        for (int l = 0; l < n; ++l) {
          dy_[l] = DY[l * LD + lane];
          lx_[l] = RAW[DM::OFF_LX + l];
        }
#pragma unroll
        for (int l = 0; l < n; ++l) v_ += dy_[l] * lx_[l];
        lxu[sl] += w * v_;
      }
      if (lane == nl - 1) KK[5 * n + sg] = RAW[DM::OFF_COST];
    });
    ex.sync();
    // DY^T Z on the Hessian tiles; lxu_i^T (w DYU) on the extra Luu tile; sum w dk
    {
      double aY[2][Exec::SLOTS][NT], bZ[2][Exec::SLOTS][NT], aX[2][Exec::SLOTS][MTU], bU[2][Exec::SLOTS][MTU];
      auto load2 = [&](int ks, int buf) {
        ex.each([&](int lane, int sl) {
          const int li = lane % 16, lq = lane / 16;
          const int kr = 4 * ks + lq;
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            aY[buf][sl][c] = DY[kr * LD + 16 * c + li];
            bZ[buf][sl][c] = Z[kr * LD + 16 * c + li];
          }
#pragma unroll
          for (int c = 0; c < MTU; ++c) {
            const int u = 16 * c + li;
            aX[buf][sl][c] = RAW[EMPC_REC_TRI ? DM::lxu(kr < n ? kr : n - 1, u < m ? u : m - 1) : DM::FULL_LXU + (kr < n ? kr : n - 1) * NM + (u < m ? u : m - 1)];
            bU[buf][sl][c] = w * DY[kr * LD + n + (u < m ? u : m - 1)];
          }
        });
      };
      load2(0, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) load2(ks + 1, (ks + 1) & 1);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int c = 0; c < NT; ++c) ex.mfma(aY[ks & 1], a, bZ[ks & 1], c, accL, a, c);
#pragma unroll
        for (int a = 0; a < MTU; ++a)
#pragma unroll
          for (int c = 0; c < MTU; ++c) ex.mfma(aX[ks & 1], a, bU[ks & 1], c, accP, a, c);
      }
    }
    ex.each([&](int lane, int sl) {
#pragma unroll
      for (int q = 0; q < SM::NSK; ++q) {
        const int e = lane + q * nl;
        if (e < n * nm) sk[sl][q] += w * DK[(e / nm) * LD + e % nm];
      }
    });
    ex.sync();
  }
  // Fx = blk(J2) (dt/6 sum w dk/dx) + blk(J1, I), Fu = blk(J2) (dt/6 sum w dk/du); the extra Luu tile -> LDS
  ex.each([&](int lane, int sl) {
    const int lj = lane % 16, lq = lane / 16;
#pragma unroll
    for (int q = 0; q < SM::NSK; ++q) {
      const int e = lane + q * nl;
      if (e < n * nm) DK[(e / nm) * LD + e % nm] = sk[sl][q] * dt / 6.0;
    }
#pragma unroll
    for (int a = 0; a < MTU; ++a)
#pragma unroll
      for (int c = 0; c < MTU; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) PT[(16 * a + 4 * r + lq) * PLD + 16 * c + lj] = accP[sl][a][c][r];
  });
  ex.sync();
  {
    const double* J1 = Jb + 3 * 72;
    const double* J2 = J1 + 36;
    const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 / 6.0 : dt / 6.0;
    ex.each([&](int lane, int sl) {
      const int lj = lane % 16, lq = lane / 16;
      if (lane < nm) {
        const int c = lane;
        double col[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = DK[i * LD + c];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double a_ = 0;
#pragma unroll
          for (int l = 0; l < 6; ++l) a_ += J2[i * 6 + l] * col[l];
          out[DM::OFF_FX + i * NM + c] = a_ + ((c < 6) ? J1[i * 6 + c] : 0.0);
        }
        for (int i = 6; i < n; ++i) out[DM::OFF_FX + i * NM + c] = DK[i * LD + c] + ((i == c) ? 1.0 : 0.0);
        out[DM::OFF_LX + c] = lxu[sl] * cscale;  // Lx | Lu are contiguous
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * a + 4 * r + lq, col = 16 * c + lj;
            const double v_ = accL[sl][a][c][r];
#if EMPC_REC_TRI
            if (row < n && col < n && DM::stored_xx(row, col)) out[DM::lxx(row, col)] = v_ * cscale;
            if (row < n && col >= n && col < nm) out[DM::lxu(row, col - n)] = v_ * cscale;
            if (row >= n && row < nm && col >= n && col < nm && DM::stored_xx(row - n, col - n))
              out[DM::luu(row - n, col - n)] = (v_ + PT[(row - n) * PLD + col - n]) * cscale;
#else
            if (row < n && col < nm) out[DM::OFF_HX + row * NM + col] = v_ * cscale;  // [Lxx Lxu], row-interleaved
            if (row >= n && row < nm && col >= n && col < nm)
              out[DM::OFF_LUU + (row - n) * m + col - n] = (v_ + PT[(row - n) * PLD + col - n]) * cscale;
#endif
          }
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
