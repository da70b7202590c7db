// TEST INFRASTRUCTURE: CPU lane emulator for the HIP kernel bodies of eagle-mpc_amd/csrc/empc_kernels.hpp.
// It executes exactly the code the GPU executes (same templates), one lane after another, so that the kernels' index
// arithmetic and staging logic can be checked against the oracle on machines without a GPU.  It is NOT a product path:
// nothing in the package loads it, and libempc.so never falls back to it.
#include <barrier>
#include <cstdio>
#include <thread>
#include <cstring>
#include <limits>
#include <vector>

#include "../../eagle-mpc_amd/csrc/empc_prep.hpp"
#include "../../eagle-mpc_amd/csrc/empc_linearize2.hpp"
#if !EMPC_REC_TRI  // (the retired generations read the full record layout: cross-checks of the default build only)
#include "superseded/empc_backward2.hpp"
#include "superseded/empc_backward3.hpp"
#endif
#include "superseded/empc_rollout5.hpp"
#include "../../eagle-mpc_amd/csrc/empc_backward4.hpp"
#include "../../eagle-mpc_amd/csrc/empc_rollout6.hpp"
#include "../../eagle-mpc_amd/csrc/empc_rk4.hpp"

using namespace empc;

template <int NL>
struct CpuExec {
  static constexpr int SLOTS = NL;
  int nl;
  template <class F>
  void each(F&& f) {
    for (int l = 0; l < nl; ++l) f(l, l);
  }
  void sync() {}
  template <class F>
  bool any(F&& f) {
    bool r = false;
    for (int l = 0; l < nl; ++l) r = f(l, l) || r;
    return r;
  }
  // EMPC_BWD_GLDS: LDS-DMA modelled as what the kernel may rely on and nothing more -- at issue the destination is POISONED (a
  // reader of the old contents, or of the new ones before the wait, meets NaN), the bytes arrive at async_wait()
  struct Pending {
    double* dst;
    const double* src;
    int n;
  };
  std::vector<Pending> pending;
  template <int ROWS>
  void async_rows(double* dst, const double* src) {
    for (int i = 0; i < 128 * ROWS; ++i) dst[i] = std::numeric_limits<double>::quiet_NaN();
    pending.push_back({dst, src, 128 * ROWS});
  }
  void async_wait() {
    for (const Pending& p : pending) std::memcpy(p.dst, p.src, sizeof(double) * p.n);
    pending.clear();
  }
  // EMPC_BWD_FUSE: wave broadcasts (v_readlane on the device)
  template <class A>
  double bcast(A& a, int j, int L) {
    return a[L][j];
  }
  template <class A>
  double bcast1(A& a, int L) {
    return a[L];
  }
  template <class A>
  bool first(A& a) {
    return a[0];
  }
  // v_mfma_f64_4x4x4_4b_f64 as documented (CDNA3 ISA guide / AMD matrix instruction calculator; tools/probes/mfma_f64_4x4_probe.hip
  // checks it on the chip): four independent products, block = (l % 16) / 4; A[block][i][k] in lane 16 k + 4 block + i,
  // B[block][k][j] in lane 16 k + 4 block + j, D[block][i][j] in lane 16 i + 4 block + j.  If the probe disagrees only these three
  // index maps (and the A fetch of the kernel) change.
  template <class A, class B, class C>
  void mfma4(A& a, int ia, B& b, int ib, C& c, int im, int in, int r) {
    double Dm[64];
    for (int l = 0; l < 64; ++l) {
      const int blk = (l % 16) / 4, i = l / 16, j = l % 4;
      double s = c[l][im][in][r];
      for (int k = 0; k < 4; ++k) s += a[16 * k + 4 * blk + i][ia] * b[16 * k + 4 * blk + j][ib];
      Dm[l] = s;
    }
    for (int l = 0; l < 64; ++l) c[l][im][in][r] = Dm[l];
  }
  // v_mfma_f64_16x16x4_f64 semantics on per-lane operands: A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16],
  // D[i = 4 r + l / 16][j = l % 16] (layout verified on gfx950 by tools/probes/mfma_f64_layout.hip)
  template <class A, class B, class C>
  void mfma(A& a, int ia, B& b, int ib, C& c, int im, int in) {
    double Dm[16][16];
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = c[(i % 4) * 16 + j][im][in][i / 4];
        for (int k = 0; k < 4; ++k) s += a[k * 16 + i][ia] * b[k * 16 + j][ib];
        Dm[i][j] = s;
      }
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) c[(i % 4) * 16 + j][im][in][i / 4] = Dm[i][j];
  }
};

// role-split kernels (warp specialisation): the four role wavefronts of a workgroup run as four host threads that meet
// at a real barrier wherever the kernel has its workgroup barrier
struct CpuRoleExec {
  static constexpr int SLOTS = 64;
  std::barrier<>* bar;
  template <class F>
  void each(F&& f) {
    for (int l = 0; l < 64; ++l) f(l, l);
  }
  void sync() { bar->arrive_and_wait(); }
};
template <class DM, int CT, bool RK4>
static void emu_rollout_group6_i(const DevBuffers& D, int grp, double* smem) {
  std::barrier<> bar(R6_WAVES);
  std::thread th[R6_WAVES];
  th[0] = std::thread([&] { CpuRoleExec ex{&bar}; rollout_group6<DM, CT, R6_A, CpuRoleExec, RK4>(ex, D, grp, smem); });
  th[1] = std::thread([&] { CpuRoleExec ex{&bar}; rollout_group6<DM, CT, R6_B, CpuRoleExec, RK4>(ex, D, grp, smem); });
  th[2] = std::thread([&] { CpuRoleExec ex{&bar}; rollout_group6<DM, CT, R6_C, CpuRoleExec, RK4>(ex, D, grp, smem); });
  th[3] = std::thread([&] { CpuRoleExec ex{&bar}; rollout_group6<DM, CT, R6_D, CpuRoleExec, RK4>(ex, D, grp, smem); });
  for (auto& t : th) t.join();
}
template <class DM, int CT>
static void emu_rollout_group6(const DevBuffers& D, int grp, double* smem) {
  if (D.integrator == EMPC_INTEGRATOR_RK4)
    emu_rollout_group6_i<DM, CT, true>(D, grp, smem);
  else
    emu_rollout_group6_i<DM, CT, false>(D, grp, smem);
}

struct Emu {
  HostProblem H;
  int B, T, NA, nb, nrot;
  std::vector<TrajState> st;
  std::vector<double> x0, xs, us, acc, tape, K, kff, Vx, Vf, xs_try, us_try, acc_try, try_cost, try_dv, us_last;
  std::vector<int> try_ok, try_ncalc, lin_knots;
  int n_lean;
  int n_active;
  int rec, nx, ndx, nu, nv;
  DevBuffers D;
  int sweeps;
};

template <class DM>
static void emu_alloc(Emu& e) {
  const int B = e.B, T = e.T, NA = e.NA;
  e.rec = DM::REC_FULL;  // what the API hands out (EMPC_REC_TRI: the device stride DM::REC is smaller)
  e.nx = DM::NX;
  e.ndx = DM::NDX;
  e.nu = DM::NU;
  e.nv = DM::NV;
  e.st.resize(B);
  e.x0.assign((size_t)B * DM::NX, 0);
  e.xs.assign((size_t)B * (T + 1) * DM::NX, 0);
  e.us.assign((size_t)B * T * DM::NU, 0);
  e.acc.assign((size_t)B * (T + 1) * DM::NACC, 0);
  e.tape.assign((size_t)B * (T + 1) * DM::REC + 128, 0);  // slack for the backward pass's whole-row prefetch
  e.K.assign((size_t)B * T * DM::NU * DM::NDX, 0);
  e.kff.assign((size_t)B * T * DM::NU, 0);
  e.Vx.assign((size_t)B * (T + 1) * DM::NDX, 0);
  e.Vf.assign((size_t)B * (T + 1) * DM::NDX, 0);
  e.xs_try.assign((size_t)B * NA * (T + 1) * DM::NX, 0);
  e.us_try.assign((size_t)B * NA * T * DM::NU, 0);
  e.acc_try.assign((size_t)B * NA * (T + 1) * DM::NACC, 0);
  e.try_cost.assign((size_t)B * NA, 0);
  e.try_dv.assign((size_t)B * NA, 0);
  e.try_ok.assign((size_t)B * NA, 0);
  e.try_ncalc.assign((size_t)B * NA, 0);
  e.us_last.assign((size_t)B * T * DM::NU, 0);
  DevBuffers& D = e.D;
  D.P = &e.H.P;
  D.sets = e.H.sets.data();
  D.set_info = e.H.set_info.data();
  D.knot_set = e.H.knot_set.data();
  D.st = e.st.data();
  D.x0 = e.x0.data();
  D.xs = e.xs.data();
  D.us = e.us.data();
  D.acc = e.acc.data();
  D.tape = e.tape.data();
  D.K = e.K.data();
  D.kff = e.kff.data();
  D.Vx = e.Vx.data();
  D.Vf = e.Vf.data();
  D.xs_try = e.xs_try.data();
  D.us_try = e.us_try.data();
  D.acc_try = e.acc_try.data();
  D.try_cost = e.try_cost.data();
  D.try_dv = e.try_dv.data();
  D.try_ok = e.try_ok.data();
  D.try_ncalc = e.try_ncalc.data();
  D.us_last = e.us_last.data();
  D.n_active = &e.n_active;
  D.integrator = e.H.P.integrator;
  D.dbg = nullptr;
  D.B = B;
  D.T = T;
  D.NA = NA;
  D.gaptol = e.H.P.prm.th_gaptol > 1e-13 ? e.H.P.prm.th_gaptol : 1e-13;
  e.n_lean = group_linearize_knots(e.H, e.lin_knots);
  D.lin_knots = e.lin_knots.data();
  D.n_lean = e.n_lean;
  for (int b = 0; b < B; ++b) std::memcpy(&e.x0[(size_t)b * DM::NX], e.H.x0.data(), sizeof(double) * DM::NX);
}

// constraint rows of the contact of knot t: the problem's, or the knot's own in a problem with stages of both types
// robot classes with a two-contact (CT_PAIR3) instantiation, as in the product (empc_solver.hip find_table)
template <class DM>
constexpr bool emu_pair_class() { return (DM::NB == 4 || DM::NB == 6) && DM::NROT == 6; }
static int emu_knot_rows(const Emu& e, int t) {
  const EmpcCostSet& set = e.H.sets[e.H.knot_set[t]];
  if (e.H.contact_rows == CT_PAIR3) return set.ncontacts > 1 ? CT_PAIR3 : 3;  // (as lin_block: the six-row body on two-contact knots)
  if (e.H.contact_rows != CT_MIXED) return e.H.contact_rows;
  return (set.ncontacts > 0 && set.contacts[0].type == EMPC_CONTACT_6D) ? 6 : 3;
}
template <class DM>
static void emu_calc(Emu& e) {
  const bool ct = e.H.P.has_contact != 0;
  for (int b = 0; b < e.B; ++b)
    for (int t = 0; t <= e.T; ++t) {
      if constexpr (true) {  // (every robot class has contact instantiations since round 4)
        if (ct) {
          if (e.H.contact_rows == CT_PAIR3) {
            if constexpr (emu_pair_class<DM>()) calc_thread<DM, CT_PAIR3>(e.D, b, t);
          } else if (e.H.contact_rows == CT_MIXED) {
            calc_thread<DM, CT_MIXED>(e.D, b, t);
          } else if (e.H.contact_rows == 6) {
            calc_thread<DM, 6>(e.D, b, t);
          } else {
            calc_thread<DM, 3>(e.D, b, t);
          }
          continue;
        }
      }
      calc_thread<DM, 0>(e.D, b, t);
    }
}
static int g_lin_version = 2;
template <class DM>
static void emu_linearize_view(Emu& e, const DevBuffers& D);
// IntegratedActionModelRK4 nodes: stage states, raw records of the stage batch, assembly (empc_rk4.hpp)
template <class DM>
static void emu_linearize_rk4(Emu& e) {
  const size_t B = e.B, T = e.T;
  std::vector<double> ys(4 * B * (T + 1) * DM::NX, 0.0), accs(4 * B * (T + 1) * DM::NACC, 0.0), us4(4 * B * T * DM::NU, 0.0),
      tape4(4 * B * (T + 1) * DM::REC + 128, 0.0);
  std::vector<TrajState> st4(4 * B);
  Rk4Buffers R{ys.data(), accs.data(), us4.data(), tape4.data(), st4.data()};
  const bool ct = e.H.P.has_contact != 0;
  for (int t = 0; t <= e.T; ++t)
    for (int b = 0; b < e.B; ++b) {
      if constexpr (true) {  // (every robot class has contact instantiations since round 4)
        if (ct) {
          if (e.H.contact_rows == CT_PAIR3) {
            if constexpr (emu_pair_class<DM>()) rk4_stage_thread<DM, CT_PAIR3>(e.D, R, b, t);
          } else if (e.H.contact_rows == CT_MIXED) {
            rk4_stage_thread<DM, CT_MIXED>(e.D, R, b, t);
          } else if (e.H.contact_rows == 6) {
            rk4_stage_thread<DM, 6>(e.D, R, b, t);
          } else {
            rk4_stage_thread<DM, 3>(e.D, R, b, t);
          }
          continue;
        }
      }
      rk4_stage_thread<DM, 0>(e.D, R, b, t);
    }
  DevBuffers Dv = e.D;
  Dv.B = 4 * e.B;
  Dv.st = st4.data();
  Dv.xs = ys.data();
  Dv.us = us4.data();
  Dv.acc = accs.data();
  Dv.tape = tape4.data();
  Dv.x0 = ys.data();
  Dv.raw = 1;
  emu_linearize_view<DM>(e, Dv);
  std::vector<double> smem(Rk4Smem<DM>::SIZE);
  for (int t = 0; t <= e.T; ++t)
    for (int b = 0; b < e.B; ++b) {
      CpuExec<64> ex{64};
      rk4_assemble_unit<DM>(ex, e.D, R, b, t, 64, smem.data());
    }
}
template <class DM>
static void emu_linearize(Emu& e) {
  if (e.H.P.integrator == EMPC_INTEGRATOR_RK4) {
    emu_linearize_rk4<DM>(e);
    return;
  }
  emu_linearize_view<DM>(e, e.D);
}
template <class DM>
static void emu_linearize_view(Emu& e, const DevBuffers& Dl) {
  constexpr int LPU = lin_lanes_per_unit<DM, 0>(), LPUC = lin_lanes_per_unit<DM, 3>();  // (the 11-dof class: 32 folded / 64 with contacts)
  std::vector<double> smem(Lin2Smem<DM>::SIZE);
  // LDS is not zeroed on the device: EMU_POISON=1 fills the unit's block with a NaN pattern (as doubles) / huge negative
  // numbers (as integers) before every unit, so that a read of something the unit did not write shows up in the results
  static const bool poison = std::getenv("EMU_POISON") != nullptr;
  for (int t = 0; t <= e.T; ++t)
    for (int b = 0; b < Dl.B; ++b) {
      const TrajState& st = Dl.st[b];
      if (st.phase == PHASE_DONE || !st.need_lin) continue;
      if (poison) std::memset(smem.data(), 0xFF, smem.size() * sizeof(double));
      if constexpr (true) {  // (every robot class has contact instantiations since round 4)
        if (e.H.P.has_contact) {
          CpuExec<64> ex{LPUC};
          constexpr int LPU = LPUC;
          if (emu_knot_rows(e, t) == CT_PAIR3) {
            if constexpr (emu_pair_class<DM>()) linearize_unit2<DM, CT_PAIR3, false>(ex, Dl, b, t, LPU, smem.data());
          } else if (emu_knot_rows(e, t) == 6) {
            linearize_unit2<DM, 6, false>(ex, Dl, b, t, LPU, smem.data());
          } else {
            linearize_unit2<DM, 3, false>(ex, Dl, b, t, LPU, smem.data());
          }
          if (emu_knot_rows(e, t) == CT_PAIR3) {
            if constexpr (emu_pair_class<DM>()) linearize_unit2<DM, CT_PAIR3, true>(ex, Dl, b, t, LPU, smem.data());
          } else if (emu_knot_rows(e, t) == 6) {
            linearize_unit2<DM, 6, true>(ex, Dl, b, t, LPU, smem.data());
          } else {
            linearize_unit2<DM, 3, true>(ex, Dl, b, t, LPU, smem.data());
          }
          continue;
        }
      }
      CpuExec<64> ex{LPU};
      linearize_unit2<DM, 0, false>(ex, Dl, b, t, LPU, smem.data());
      if (poison) std::memset(smem.data(), 0xFF, smem.size() * sizeof(double));
      linearize_unit2<DM, 0, true>(ex, Dl, b, t, LPU, smem.data());
    }
}
static int g_bwd_version = 2;
template <class DM>
static void emu_backward(Emu& e) {
#if !EMPC_REC_TRI
  std::vector<double> smem(Bwd2Smem<DM>::SIZE);
  std::vector<double> smem3(Bwd3Smem<DM>::SIZE);
#endif
  std::vector<double> smem4(Bwd4Smem<DM>::SIZE);
  for (int b = 0; b < e.B; ++b) {
    if (g_bwd_version == 4 || e.H.P.prm.solver_type != EMPC_SOLVER_SBFDDP || EMPC_REC_TRI) {
      CpuExec<64> ex{64};
      if (e.H.P.prm.solver_type != EMPC_SOLVER_SBFDDP)
        backward_traj4<DM, true>(ex, e.D, b, smem4.data());
      else
        backward_traj4<DM, false>(ex, e.D, b, smem4.data());  // the shipped form: matrix cores, zero-padded tiles
      continue;
    }
#if !EMPC_REC_TRI
    if (g_bwd_version == 3) {
      CpuExec<64> ex{64};
      backward_traj3<DM>(ex, e.D, b, smem3.data());  // the shipped matrix-core form
      continue;
    }
    if (g_bwd_version == 2) {
      CpuExec<256> ex{256};
      backward_traj2<DM, 256>(ex, e.D, b, smem.data());  // four-wavefront tiling
    } else {
      CpuExec<256> ex{64};
      backward_traj2<DM, 64>(ex, e.D, b, smem.data());   // the shipped single-wavefront configuration
    }
#endif
  }
}
static int g_roll_version = 6;
template <class DM>
static void emu_rollout(Emu& e) {
  const bool ct = e.H.P.has_contact != 0;
  if ((g_roll_version == 6 || (g_roll_version == 5 && e.H.P.prm.solver_type != EMPC_SOLVER_SBFDDP)) && e.NA <= MAX_ALPHAS) {
    // the shipped form: packed trajectories, role wavefronts (Euler nodes, and RK4 nodes as four stages per knot)
    const int G = roll6_group_size(e.NA);
    std::vector<double> smem6(Roll6Smem<DM>::size_for(CT_PAIR3));
    for (int grp = 0; grp * G < e.B; ++grp) {
      if constexpr (true) {  // (every robot class has contact instantiations since round 4)
        if (ct) {
          if (e.H.contact_rows == CT_PAIR3) {
            if constexpr (emu_pair_class<DM>()) emu_rollout_group6<DM, CT_PAIR3>(e.D, grp, smem6.data());
          } else if (e.H.contact_rows == CT_MIXED) {
            emu_rollout_group6<DM, CT_MIXED>(e.D, grp, smem6.data());
          } else if (e.H.contact_rows == 6) {
            emu_rollout_group6<DM, 6>(e.D, grp, smem6.data());
          } else {
            emu_rollout_group6<DM, 3>(e.D, grp, smem6.data());
          }
          continue;
        }
      }
      emu_rollout_group6<DM, 0>(e.D, grp, smem6.data());
    }
    return;
  }
  for (int b = 0; b < e.B; ++b)
    for (int ai = 0; ai < e.NA; ++ai) {
      if (g_roll_version >= 5 && e.H.P.integrator == EMPC_INTEGRATOR_EULER && e.H.contact_rows != CT_PAIR3) {  // (the retired wave form has no two-contact body)
        if (ai > 0) continue;  // one call per trajectory: the 64 lanes cover every step length
        std::vector<double> smem5(Roll5Smem<DM>::SIZE);
        CpuExec<64> ex{64};
        if constexpr (true) {  // (every robot class has contact instantiations since round 4)
          if (ct) {
            if (e.H.contact_rows == CT_MIXED) {
              rollout_wave5<DM, CT_MIXED>(ex, e.D, b, 64, smem5.data());
            } else if (e.H.contact_rows == 6) {
              rollout_wave5<DM, 6>(ex, e.D, b, 64, smem5.data());
            } else {
              rollout_wave5<DM, 3>(ex, e.D, b, 64, smem5.data());
            }
            continue;
          }
        }
        rollout_wave5<DM, 0>(ex, e.D, b, 64, smem5.data());
        continue;
      }
      if constexpr (true) {  // (every robot class has contact instantiations since round 4)
        if (ct) {
          if (e.H.contact_rows == CT_PAIR3) {
            if constexpr (emu_pair_class<DM>()) rollout_thread<DM, CT_PAIR3>(e.D, b, ai);
          } else if (e.H.contact_rows == CT_MIXED) {
            rollout_thread<DM, CT_MIXED>(e.D, b, ai);
          } else if (e.H.contact_rows == 6) {
            rollout_thread<DM, 6>(e.D, b, ai);
          } else {
            rollout_thread<DM, 3>(e.D, b, ai);
          }
          continue;
        }
      }
      rollout_thread<DM, 0>(e.D, b, ai);
    }
}
template <class DM>
static void emu_select(Emu& e) {
  e.n_active = 0;
  for (int b = 0; b < e.B; ++b) {
    int acc_ai, last_ai;
    select_decide<DM>(e.D, b, acc_ai, last_ai);
    select_copy<DM>(e.D, b, acc_ai, last_ai, 0, 1);
    if (e.D.q_rows) {  // streamed solves: the hand-over k_select performs
      TrajState& st = e.st[b];
      if (st.phase == PHASE_DONE && st.job >= 0) {
        stream_write_row<DM>(e.D, b, 0, 1);
        *e.D.q_iters += (unsigned long long)st.total_iters;
        const int j = (*e.D.q_head)++;
        const int job = j < e.D.q_njobs ? j : -1;
        if (job >= 0) {
          stream_refill<DM>(e.D, b, job, 0, 1);
          traj_state_init(st, e.H.P.prm, e.D.q_maxiter, false, (const TrajState*)nullptr);
        }
        st.job = job;
      }
    }
    if (e.st[b].phase != PHASE_DONE) e.n_active++;
  }
}
template <class DM>
static void emu_run_sweeps(Emu& e) {
  e.sweeps = 0;
  while (true) {
    emu_calc<DM>(e);
    emu_linearize<DM>(e);
    emu_backward<DM>(e);
    emu_rollout<DM>(e);
    emu_select<DM>(e);
    e.sweeps++;
    if (e.n_active == 0 || e.sweeps > 100000) break;
  }
}
template <class DM>
static void emu_stream(Emu& e, int n_jobs, const double* x0s, int maxiter, double* rows, long long* total_iters) {
  const int nfirst = n_jobs < e.B ? n_jobs : e.B;
  int head = nfirst;
  unsigned long long iters = 0;
  std::fill(e.xs.begin(), e.xs.end(), 0.0);
  for (size_t i = 0; i < (size_t)e.B * (e.T + 1); ++i) e.xs[i * e.nx + 6] = 1.0;
  std::fill(e.us.begin(), e.us.end(), 0.0);
  std::fill(e.kff.begin(), e.kff.end(), 0.0);
  std::memcpy(e.x0.data(), x0s, sizeof(double) * (size_t)nfirst * e.nx);
  for (int b = 0; b < e.B; ++b) {
    init_traj_state(e.st[b], e.H.P.prm, maxiter, false, nullptr);
    e.st[b].job = b < nfirst ? b : -1;
    if (b >= nfirst) e.st[b].phase = PHASE_DONE;
  }
  const DevBuffers keep = e.D;
  e.D.q_x0 = x0s;
  e.D.q_rows = rows;
  e.D.q_head = &head;
  e.D.q_iters = &iters;
  e.D.q_njobs = n_jobs;
  e.D.q_maxiter = maxiter;
  emu_run_sweeps<DM>(e);
  e.D = keep;
  if (total_iters) *total_iters = (long long)iters;
}
template <class DM>
static void emu_sweep_stages(Emu& e, int stages) {
  if (stages & EMPC_STAGE_LINEARIZE) {
    emu_calc<DM>(e);
    emu_linearize<DM>(e);
  }
  if (stages & EMPC_STAGE_BACKWARD) emu_backward<DM>(e);
  if (stages & EMPC_STAGE_ROLLOUT) emu_rollout<DM>(e);
  if (stages & EMPC_STAGE_SELECT) emu_select<DM>(e);
}
template <class DM>
static int emu_row_doubles(Emu& e) {
  return (int)stream_row_doubles<DM>(e.T);
}
template <class DM>
static void emu_solve(Emu& e, int maxiter, int is_feasible) {
  for (int b = 0; b < e.B; ++b) {
    TrajState prev = e.st[b];
    init_traj_state(e.st[b], e.H.P.prm, maxiter, is_feasible != 0, &prev);
  }
  if (e.H.P.prm.solver_type != EMPC_SOLVER_SBFDDP) std::fill(e.kff.begin(), e.kff.end(), 0.0);
  emu_run_sweeps<DM>(e);
}

#ifdef EMU_BAKED
// -DEMU_BAKED: the kernel bodies instantiated over the BAKED robot tables (csrc/baked/empc_baked_models.hpp), the family the
// shipped library runs for every shipped robot.  emu_create refuses a robot that differs from its table by one bit.  A build of
// its own (tests/test_emulator_parity.py::test_baked_family_equals_runtime_family builds it on demand): the default emulator
// keeps the runtime-model family and its compile time.
template <class M>
static bool emu_baked_is(const Emu* e) {
  const BakedTree& t = M::tree();
  const EmpcModelDesc& m = e->H.P.model;
  if (m.nbodies != t.nbodies || m.nq != t.nq || m.nv != t.nv || e->H.P.n_rotors != t.n_rotors) return false;
  auto same = [](const double* a, const double* b, int n) { return std::memcmp(a, b, sizeof(double) * n) == 0; };
  for (int b = 0; b < m.nbodies; ++b) {
    if (m.parent[b] != t.parent[b]) return false;
    if (!same(m.jplace_R[b], t.jplace_R[b], 9) || !same(m.jplace_p[b], t.jplace_p[b], 3) || !same(m.axis[b], t.axis[b], 3) ||
        !same(&m.mass[b], &t.mass[b], 1) || !same(m.com[b], t.com[b], 3) || !same(m.inertia[b], t.inertia[b], 9))
      return false;
  }
  return same(m.gravity, t.gravity, 3) && same(e->H.P.tau_f, t.tau_f, 6 * e->H.P.n_rotors) && same(e->H.P.u_lb, t.u_lb, e->H.P.nu) &&
         same(e->H.P.u_ub, t.u_ub, e->H.P.nu);
}
#define DISPATCH(e, FN, ...)                                                                         \
  do {                                                                                               \
    if (emu_baked_is<BakedHex370Arm3>(e)) FN<Dims<4, 6, BakedHex370Arm3>>(__VA_ARGS__);              \
    else if (emu_baked_is<BakedHextiltArm5>(e)) FN<Dims<6, 6, BakedHextiltArm5>>(__VA_ARGS__);       \
    else if (emu_baked_is<BakedHex370>(e)) FN<Dims<1, 6, BakedHex370>>(__VA_ARGS__);                 \
    else { std::fprintf(stderr, "emulator (baked build): robot is not one of the baked tables\n"); } \
  } while (0)
#else
#define DISPATCH(e, FN, ...)                                                    \
  do {                                                                          \
    if (e->nb == 1 && e->nrot == 6) FN<Dims<1, 6>>(__VA_ARGS__);                \
    else if (e->nb == 1 && e->nrot == 4) FN<Dims<1, 4>>(__VA_ARGS__);           \
    else if (e->nb == 3 && e->nrot == 6) FN<Dims<3, 6>>(__VA_ARGS__);           \
    else if (e->nb == 4 && e->nrot == 6) FN<Dims<4, 6>>(__VA_ARGS__);           \
    else if (e->nb == 6 && e->nrot == 6) FN<Dims<6, 6>>(__VA_ARGS__);           \
    else { std::fprintf(stderr, "emulator: unsupported dims\n"); }             \
  } while (0)
#endif

template <class DM>
static void emu_node(Emu& e, int t, const double* x, const double* u, double smooth, double* xnext, double* acc, double* cost,
                     double* usq, double* lam) {
  const EmpcCostSet& set = e.H.sets[e.H.knot_set[t]];
  double c = 0;
  if constexpr (true) {  // (every robot class has contact instantiations since round 4)
    if (e.H.P.has_contact) {
      if (e.H.contact_rows == CT_PAIR3) {
        if constexpr (emu_pair_class<DM>()) node_nominal<DM, CT_PAIR3>(e.H.P, set, smooth, x, u, u == nullptr, xnext, acc, c, usq, lam);
      } else if (e.H.contact_rows == CT_MIXED) {
        node_nominal<DM, CT_MIXED>(e.H.P, set, smooth, x, u, u == nullptr, xnext, acc, c, usq, lam);
      } else if (e.H.contact_rows == 6) {
        node_nominal<DM, 6>(e.H.P, set, smooth, x, u, u == nullptr, xnext, acc, c, usq, lam);
      } else {
        node_nominal<DM, 3>(e.H.P, set, smooth, x, u, u == nullptr, xnext, acc, c, usq, lam);
      }
      *cost = c;
      return;
    }
  }
  node_nominal<DM, 0>(e.H.P, set, smooth, x, u, u == nullptr, xnext, acc, c, usq, lam);
  *cost = c;
}

template <class DM>
static void emu_tape_out(Emu& e, double* tape) {
  const size_t nrec = (size_t)e.B * (e.T + 1);
  for (size_t i = 0; i < nrec; ++i) unpack_record<DM>(e.tape.data() + i * DM::REC, tape + i * DM::REC_FULL);
}
extern "C" {
// IAM.calc of one node through the device code path (node_nominal: Euler or RK4 as the problem says)
void emu_node_nominal(void* h, int t, const double* x, const double* u, double smooth, double* xnext, double* acc, double* cost,
                      double* usq, double* lam) {
  Emu* e = static_cast<Emu*>(h);
  DISPATCH(e, emu_node, *e, t, x, u, smooth, xnext, acc, cost, usq, lam);
}
void emu_set_linearize_version(int v) { g_lin_version = v; }
void emu_set_backward_version(int v) { g_bwd_version = v; }
void emu_set_rollout_version(int v) { g_roll_version = v; }

void* emu_create(const EmpcProblemDesc* d, const EmpcSolverParams* prm, int B) {
  Emu* e = new Emu();
  try {
    prepare_problem(*d, *prm, e->H);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "emu_create: %s\n", ex.what());
    delete e;
    return nullptr;
  }
  e->B = B;
  e->T = d->T;
  e->NA = prm->n_alphas;
  e->nb = d->model.nbodies;
  e->nrot = d->n_rotors;
#ifdef EMU_BAKED
  if (!(emu_baked_is<BakedHex370Arm3>(e) || emu_baked_is<BakedHextiltArm5>(e) || emu_baked_is<BakedHex370>(e))) {
    std::fprintf(stderr, "emu_create (baked build): robot is not one of the baked tables\n");
    delete e;
    return nullptr;
  }
#endif
  DISPATCH(e, emu_alloc, *e);
  std::memset(e->st.data(), 0, sizeof(TrajState) * B);
  return e;
}
void emu_destroy(void* h) { delete static_cast<Emu*>(h); }
int emu_rec(void* h) { return static_cast<Emu*>(h)->rec; }
void emu_set_x0(void* h, const double* x0s) {
  Emu* e = static_cast<Emu*>(h);
  std::memcpy(e->x0.data(), x0s, sizeof(double) * e->x0.size());
}
void emu_set_warmstart(void* h, const double* xs, const double* us) {
  Emu* e = static_cast<Emu*>(h);
  if (xs)
    std::memcpy(e->xs.data(), xs, sizeof(double) * e->xs.size());
  else {
    std::fill(e->xs.begin(), e->xs.end(), 0.0);
    for (size_t i = 0; i < (size_t)e->B * (e->T + 1); ++i) e->xs[i * e->nx + 6] = 1.0;
  }
  if (us)
    std::memcpy(e->us.data(), us, sizeof(double) * e->us.size());
  else
    std::fill(e->us.begin(), e->us.end(), 0.0);
}
int emu_solve_c(void* h, int maxiter, int is_feasible) {
  Emu* e = static_cast<Emu*>(h);
  DISPATCH(e, emu_solve, *e, maxiter, is_feasible);
  return e->sweeps;
}
void emu_get(void* h, double* xs, double* us, double* us_last, double* cost, int* iters, int* status) {
  Emu* e = static_cast<Emu*>(h);
  if (xs) std::memcpy(xs, e->xs.data(), sizeof(double) * e->xs.size());
  if (us) std::memcpy(us, e->us.data(), sizeof(double) * e->us.size());
  if (us_last) std::memcpy(us_last, e->us_last.data(), sizeof(double) * e->us_last.size());
  for (int b = 0; b < e->B; ++b) {
    if (cost) cost[b] = e->st[b].cost;
    if (iters) iters[b] = e->st[b].iter;
    if (status) status[b] = e->st[b].status;
  }
}
// phase level: state is set up the way the solver would have it inside a pass
void emu_phase_setup(void* h, double smooth, int is_feasible, double xreg, int ddp) {
  Emu* e = static_cast<Emu*>(h);
  for (int b = 0; b < e->B; ++b) {
    TrajState& s = e->st[b];
    init_traj_state(s, e->H.P.prm, 100, false, nullptr);
    s.smooth = smooth;
    s.is_feasible = is_feasible;
    s.xreg = s.ureg = xreg;
    s.phase = ddp ? PHASE_DDP : 0;
  }
}
void emu_phase_linearize(void* h, double* tape, double* acc) {
  Emu* e = static_cast<Emu*>(h);
  DISPATCH(e, emu_calc, *e);
  DISPATCH(e, emu_linearize, *e);
  if (tape) DISPATCH(e, emu_tape_out, *e, tape);
  if (acc) {
    const int nacc = e->nv + 6;
    for (size_t u = 0; u < (size_t)e->B * (e->T + 1); ++u) std::memcpy(acc + u * e->nv, &e->acc[u * nacc], sizeof(double) * e->nv);
  }
}
void emu_phase_backward(void* h, double* K, double* k, double* Vx, double* dgdq, int* ok, int* feas, double* cost) {
  Emu* e = static_cast<Emu*>(h);
  DISPATCH(e, emu_backward, *e);
  if (K) std::memcpy(K, e->K.data(), sizeof(double) * e->K.size());
  if (k) std::memcpy(k, e->kff.data(), sizeof(double) * e->kff.size());
  if (Vx) std::memcpy(Vx, e->Vx.data(), sizeof(double) * e->Vx.size());
  for (int b = 0; b < e->B; ++b) {
    const TrajState& s = e->st[b];
    if (dgdq) {
      dgdq[2 * b] = s.dg_u + (s.is_feasible ? 0.0 : s.dg_f);
      dgdq[2 * b + 1] = s.dq_u + (s.is_feasible ? 0.0 : s.dq_f);
    }
    if (ok) ok[b] = !s.bwd_failed;
    if (feas) feas[b] = s.is_feasible;
    if (cost) cost[b] = s.cost;
  }
}
void emu_phase_rollout(void* h, int ai, double* xs_try, double* us_try, double* cost_try, double* dv, int* ok) {
  Emu* e = static_cast<Emu*>(h);
  for (int b = 0; b < e->B; ++b) e->st[b].need_lin = 0;
  DISPATCH(e, emu_rollout, *e);
  const int T = e->T, NA = e->NA;
  for (int b = 0; b < e->B; ++b) {
    const size_t slot = (size_t)b * NA + ai;
    if (xs_try) std::memcpy(xs_try + (size_t)b * (T + 1) * e->nx, &e->xs_try[slot * (T + 1) * e->nx], sizeof(double) * (T + 1) * e->nx);
    if (us_try) std::memcpy(us_try + (size_t)b * T * e->nu, &e->us_try[slot * T * e->nu], sizeof(double) * T * e->nu);
    if (cost_try) cost_try[b] = e->try_cost[slot];
    if (dv) dv[b] = e->try_dv[slot];
    if (ok) ok[b] = e->try_ok[slot];
  }
}

// ---- step-wise entry points (mirror of empc_solver_get_states / set_states / empc_sweep_batch / empc_select_batch) ----
void emu_get_states(void* h, EmpcTrajState* out) {
  Emu* e = static_cast<Emu*>(h);
  std::memcpy(out, e->st.data(), sizeof(TrajState) * e->B);
}
void emu_set_states(void* h, const EmpcTrajState* in) {
  Emu* e = static_cast<Emu*>(h);
  std::memcpy(e->st.data(), in, sizeof(TrajState) * e->B);
}
void emu_sweep(void* h, int stages) {
  Emu* e = static_cast<Emu*>(h);
  DISPATCH(e, emu_sweep_stages, *e, stages);
}
void emu_set_trials(void* h, const int* ok, const double* cost, const double* dv) {
  Emu* e = static_cast<Emu*>(h);
  const size_t n = (size_t)e->B * e->NA;
  if (ok) std::memcpy(e->try_ok.data(), ok, sizeof(int) * n);
  if (cost) std::memcpy(e->try_cost.data(), cost, sizeof(double) * n);
  if (dv) std::memcpy(e->try_dv.data(), dv, sizeof(double) * n);
}
void emu_get_trials(void* h, double* cost, double* dv, int* ok) {
  Emu* e = static_cast<Emu*>(h);
  const size_t n = (size_t)e->B * e->NA;
  if (cost) std::memcpy(cost, e->try_cost.data(), sizeof(double) * n);
  if (dv) std::memcpy(dv, e->try_dv.data(), sizeof(double) * n);
  if (ok) std::memcpy(ok, e->try_ok.data(), sizeof(int) * n);
}
void emu_get_tape(void* h, double* tape) {
  Emu* e = static_cast<Emu*>(h);
  DISPATCH(e, emu_tape_out, *e, tape);
}
void emu_get_gains(void* h, double* K, double* k, double* Vx) {
  Emu* e = static_cast<Emu*>(h);
  if (K) std::memcpy(K, e->K.data(), sizeof(double) * e->K.size());
  if (k) std::memcpy(k, e->kff.data(), sizeof(double) * e->kff.size());
  if (Vx) std::memcpy(Vx, e->Vx.data(), sizeof(double) * e->Vx.size());
}
void emu_set_gains(void* h, const double* K, const double* k) {
  Emu* e = static_cast<Emu*>(h);
  if (K) std::memcpy(e->K.data(), K, sizeof(double) * e->K.size());
  if (k) std::memcpy(e->kff.data(), k, sizeof(double) * e->kff.size());
}
// streamed solves: n_jobs solves through the emulator's B slots; rows as empc_solver_stream_results
int emu_stream_row_doubles(void* h) {
  Emu* e = static_cast<Emu*>(h);
  int n = 0;
  // (the row layout depends on the dimensions only, not on the model family)
  if (e->nb == 1 && e->nrot == 6) n = emu_row_doubles<Dims<1, 6>>(*e);
  else if (e->nb == 1 && e->nrot == 4) n = emu_row_doubles<Dims<1, 4>>(*e);
  else if (e->nb == 3 && e->nrot == 6) n = emu_row_doubles<Dims<3, 6>>(*e);
  else if (e->nb == 4 && e->nrot == 6) n = emu_row_doubles<Dims<4, 6>>(*e);
  else if (e->nb == 6 && e->nrot == 6) n = emu_row_doubles<Dims<6, 6>>(*e);
  return n;
}
int emu_stream_c(void* h, int n_jobs, const double* x0s, int maxiter, double* rows, long long* total_iters) {
  Emu* e = static_cast<Emu*>(h);
  DISPATCH(e, emu_stream, *e, n_jobs, x0s, maxiter, rows, total_iters);
  return e->sweeps;
}
}
