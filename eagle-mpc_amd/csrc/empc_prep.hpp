// empc_prep.hpp -- host-side preparation of the device problem image (pure C++, no HIP).
//   * copies the flat descriptor into DevProblem + cost-set table + knot table
//   * SolverSbFDDP::barrierInit (src/sbfddp.cpp:169-190): adds the "barrier" cost to the cost table of every distinct
//     running model, keeping the table in std::map (alphabetical) order
//   * validates that the problem fits what the kernels are built for
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "empc_kernels.hpp"

namespace empc {

struct HostProblem {
  DevProblem P;
  std::vector<EmpcCostSet> sets;
  std::vector<SetInfo> set_info;
  std::vector<int> knot_set;
  std::vector<double> x0;
  int contact_rows = 0;  // 0: no contact stage; 3: ContactModel3D; 6: ContactModel6D (selects the kernel instantiation)
};

inline void insert_barrier(EmpcCostSet& s, const DevProblem& P) {
  for (int i = 0; i < s.ncosts; ++i)
    if (std::strcmp(s.costs[i].name, "barrier") == 0) return;
  if (s.ncosts >= EMPC_MAX_COSTS) throw std::runtime_error("no room for the barrier cost in a cost set");
  EmpcCost c;
  std::memset(&c, 0, sizeof(c));
  std::strcpy(c.name, "barrier");
  c.type = EMPC_COST_CONTROL;
  c.activation = EMPC_ACT_WEIGHTED_QUADRATIC_BARRIER;
  c.active = 1;
  c.frame = -1;
  c.nr = P.nu;
  c.is_barrier = 1;
  c.weight = P.prm.barrier_weight;
  for (int i = 0; i < P.nu; ++i) {
    c.lb[i] = P.u_lb[i];  // s_lb = u_lb, s_ub = u_ub (SquashingModelSmoothSat)
    c.ub[i] = P.u_ub[i];
    c.act_w[i] = 1.0;     // replaced on the device by 1/(smooth (ub-lb))^2 of the trajectory's current pass
  }
  int pos = 0;
  while (pos < s.ncosts && std::strcmp(s.costs[pos].name, "barrier") < 0) ++pos;
  for (int i = s.ncosts; i > pos; --i) s.costs[i] = s.costs[i - 1];
  s.costs[pos] = c;
  s.ncosts++;
}

inline void prepare_problem(const EmpcProblemDesc& d, const EmpcSolverParams& prm, HostProblem& H) {
  if (!d.sets || !d.knot_set) throw std::invalid_argument("problem descriptor has no cost-set / knot tables");
  if (d.T < 1) throw std::invalid_argument("problem needs at least one running knot");
  if (d.integrator != EMPC_INTEGRATOR_EULER && d.integrator != EMPC_INTEGRATOR_RK4)
    throw std::invalid_argument("unknown integrator (IntegratedActionModelEuler or IntegratedActionModelRK4)");
  const EmpcModelDesc& m = d.model;
  for (int b = 1; b < m.nbodies; ++b)
    if (m.parent[b] != b - 1) throw std::runtime_error("the device kernels need a serial kinematic chain");
  if (d.nu != d.n_rotors + m.nv - 6 || d.nx != m.nq + m.nv || d.ndx != 2 * m.nv)
    throw std::invalid_argument("inconsistent problem dimensions");
  if (d.nu > m.nv) throw std::runtime_error("more controls than velocity dimensions is not supported by the kernels");
  if (prm.n_alphas < 1 || prm.n_alphas > MAX_ALPHAS) throw std::invalid_argument("n_alphas out of range");
  std::memset(&H.P, 0, sizeof(H.P));
  H.contact_rows = 0;
  H.P.model = m;
  H.P.nx = d.nx;
  H.P.ndx = d.ndx;
  H.P.nu = d.nu;
  H.P.n_rotors = d.n_rotors;
  H.P.T = d.T;
  H.P.n_sets = d.n_sets;
  H.P.has_contact = d.has_contact;
  H.P.use_squash = d.use_squash;
  H.P.integrator = d.integrator;
  H.P.dt = d.dt;
  std::memcpy(H.P.tau_f, d.tau_f, sizeof(d.tau_f));
  std::memcpy(H.P.u_lb, d.u_lb, sizeof(d.u_lb));
  std::memcpy(H.P.u_ub, d.u_ub, sizeof(d.u_ub));
  H.P.prm = prm;
  H.sets.assign(d.sets, d.sets + d.n_sets);
  H.knot_set.assign(d.knot_set, d.knot_set + d.T + 1);
  H.x0.assign(d.x0, d.x0 + d.nx);
  for (int t = 0; t <= d.T; ++t)
    if (H.knot_set[t] < 0 || H.knot_set[t] >= d.n_sets) throw std::invalid_argument("knot table references a missing cost set");
  // barrierInit: once per distinct running model -- SolverSbFDDP's own cost; the crocoddyl box solvers add nothing
  if (prm.solver_type == EMPC_SOLVER_SBFDDP) {
    std::vector<char> done(H.sets.size(), 0);
    for (int t = 0; t < d.T; ++t) {
      const int si = H.knot_set[t];
      if (done[si]) continue;
      done[si] = 1;
      insert_barrier(H.sets[si], H.P);
    }
  }
  // State costs that share a reference reuse the residual of the first one (reg_state / limits_state pairs)
  for (auto& s : H.sets)
    for (int i = 0; i < s.ncosts; ++i) {
      s.costs[i].ref_share = -1;
      if (s.costs[i].type != EMPC_COST_STATE) continue;
      for (int j = 0; j < i; ++j)
        if (s.costs[j].type == EMPC_COST_STATE && s.costs[j].active &&
            std::memcmp(s.costs[j].ref, s.costs[i].ref, sizeof(double) * d.nx) == 0) {
          s.costs[i].ref_share = j;
          break;
        }
    }
  // friction-cone costs: the 5 x 3 matrix A R_n^T depends only on the cost's normal and mu; computed here once and kept
  // in the cost's reference payload (ref[4..18]) so that the kernels do not rebuild it (sqrt / atan2 / sin / cos) per node
  for (auto& s : H.sets)
    for (int i = 0; i < s.ncosts; ++i) {
      EmpcCost& c = s.costs[i];
      if (c.type != EMPC_COST_CONTACT_FRICTION_CONE) continue;
      double AR[5][3];
      const double nsf[3] = {c.ref[0], c.ref[1], c.ref[2]};
      cone_rows(nsf, c.ref[3], AR);
      for (int r = 0; r < 5; ++r)
        for (int j = 0; j < 3; ++j) c.ref[4 + 3 * r + j] = AR[r][j];
    }
  // frame-capture capacity per cost set
  bool pair_seen = false;
  for (auto& s : H.sets) {
    int frames[EMPC_MAX_COSTS], nf = 0;
    for (int i = 0; i < s.ncosts; ++i) {
      const EmpcCost& c = s.costs[i];
      if (!c.active || c.frame < 0 || c.type == EMPC_COST_CONTACT_FRICTION_CONE) continue;
      if (c.frame >= m.nframes) throw std::invalid_argument("cost references a frame outside the model's frame table");
      bool seen = false;
      for (int k = 0; k < nf; ++k) seen = seen || frames[k] == c.frame;
      if (!seen) frames[nf++] = c.frame;
    }
    if (s.ncosts > 0) {  // flag read by linearize (set_uses_frames)
      EmpcCostSet& ms = s;
      ms.costs[0].reserved = (ms.costs[0].reserved & ~1) | ((nf > 0 || s.ncontacts >= 1) ? 1 : 0);
    }
    for (int k = 0; k < s.ncontacts && k < EMPC_MAX_CONTACTS; ++k) {  // the contact frames are captured too
      if (s.contacts[k].frame < 0 || s.contacts[k].frame >= m.nframes)
        throw std::invalid_argument("contact references a frame outside the model's frame table");
      bool seen = false;
      for (int q = 0; q < nf; ++q) seen = seen || frames[q] == s.contacts[k].frame;
      if (!seen) frames[nf++] = s.contacts[k].frame;
    }
    if (nf > NCAP) throw std::runtime_error("a cost set references more distinct frames than the kernels capture");
    if (s.ncontacts > 2) throw std::runtime_error("more than two contacts per stage are not supported by the kernels");
    if (s.ncontacts == 2) {
      // ContactModelMultiple with two entries (src/stage.cpp:38-48): the kernels have the six-row form for two ContactModel3D
      // (CT_PAIR3).  3 + 6 or 6 + 6 rows would leave the 9- and 11-dof robots of this path 0 to 2 degrees of freedom (on
      // the 9-dof arm Jc M^-1 Jc^T of a 6D + 3D pair is singular: the oracle's Cholesky fails as crocoddyl's would).
      if (s.contacts[0].type != EMPC_CONTACT_3D || s.contacts[1].type != EMPC_CONTACT_3D)
        throw std::runtime_error("two contacts per stage are supported for two ContactModel3D only (six constraint rows)");
      pair_seen = true;
    }
    if (s.ncontacts >= 1) {
      // the number of constraint rows is a compile-time constant of the kernels (3: ContactModel3D, 6: ContactModel6D)
      const int rows = s.contacts[0].type == EMPC_CONTACT_3D ? 3 : (s.contacts[0].type == EMPC_CONTACT_6D ? 6 : -1);
      if (rows < 0) throw std::runtime_error("unknown contact type in a cost set");
      // stages of both types in one problem: the mixed instantiation (CT_MIXED), which branches on the node's contact type
      H.contact_rows = (H.contact_rows != 0 && H.contact_rows != rows) ? CT_MIXED : rows;
    }
    // a ContactFrictionCone cost of a stage with two contacts reads the force of the contact on ITS frame (crocoddyl looks the
    // contact up by frame id and fails without one)
    if (s.ncontacts == 2)
      for (int i = 0; i < s.ncosts; ++i) {
        const EmpcCost& c = s.costs[i];
        if (!c.active || c.type != EMPC_COST_CONTACT_FRICTION_CONE) continue;
        if (c.frame != s.contacts[0].frame && c.frame != s.contacts[1].frame)
          throw std::runtime_error("a ContactFrictionCone cost of a stage with two contacts names a frame that carries neither");
      }
  }
  if (pair_seen) {
    // every other contact stage of the problem must be a single ContactModel3D (the three-row body behind a uniform branch)
    if (H.contact_rows != 3) throw std::runtime_error("a problem with a two-contact stage may only hold ContactModel3D contacts");
    H.contact_rows = CT_PAIR3;
  }
  // work lists per cost set (SetInfo); the capture order is the one node_nominal derives by scanning the table
  H.set_info.assign(H.sets.size(), SetInfo());
  for (size_t k = 0; k < H.sets.size(); ++k) {
    const EmpcCostSet& s = H.sets[k];
    SetInfo& I = H.set_info[k];
    std::memset(&I, 0, sizeof(I));
    for (int i = 0; i < s.ncosts; ++i) {
      const EmpcCost& c = s.costs[i];
      if (!c.active) continue;
      if (c.type == EMPC_COST_STATE) {
        I.state_ci[I.n_state++] = i;
        I.sc_ci[I.n_sc++] = i;
      } else if (c.type == EMPC_COST_CONTROL) {
        I.ctrl_ci[I.n_ctrl++] = i;
        I.sc_ci[I.n_sc++] = i;
      } else if (c.type == EMPC_COST_CONTACT_FRICTION_CONE) {
        I.cone_ci[I.n_cone++] = i;
      } else {
        I.frame_ci[I.n_frame++] = i;
      }
      if (c.frame >= 0 && c.type != EMPC_COST_CONTACT_FRICTION_CONE) {
        bool seen = false;
        for (int q = 0; q < I.ncap; ++q) seen = seen || I.capf[q] == c.frame;
        if (!seen && I.ncap < NCAP) I.capf[I.ncap++] = c.frame;
      }
    }
    for (int a = 0; a < I.n_state; ++a) {  // State costs that share a reference reuse the first one's residual (linearize)
      I.state_own[a] = a;
      for (int b = 0; b < a; ++b)
        if (std::memcmp(s.costs[I.state_ci[b]].ref, s.costs[I.state_ci[a]].ref, sizeof(double) * d.nx) == 0) {
          I.state_own[a] = b;
          break;
        }
    }
    if (d.has_contact && s.ncontacts > 0) {
      const int cframe = s.contacts[0].frame;
      bool seen = false;
      for (int q = 0; q < I.ncap; ++q)
        if (I.capf[q] == cframe) {
          seen = true;
          I.ccap = q;
        }
      if (!seen) {
        I.ccap = I.ncap;
        if (I.ncap < NCAP) I.capf[I.ncap++] = cframe;
      }
      if (s.ncontacts > 1) {  // CT_PAIR3: slot of the second contact frame in bits 8.. of ccap (read by the pair kernels only)
        const int cframe2 = s.contacts[1].frame;
        int slot2 = -1;
        for (int q = 0; q < I.ncap; ++q)
          if (I.capf[q] == cframe2) slot2 = q;
        if (slot2 < 0) {
          slot2 = I.ncap;
          if (I.ncap < NCAP) I.capf[I.ncap++] = cframe2;
        }
        I.ccap |= slot2 << 8;
      }
    }
  }
}

}  // namespace empc

namespace empc {
// State of one trajectory at the start of SolverSbFDDP::solve (src/sbfddp.cpp:198-210).  `prev` carries the members
// that the reference keeps across solve() calls (cost_, cost_prev_, stop_).
// knots grouped for linearize: the lean ones (cost set without operational frames, flag written by prepare_problem)
// first, the others after; returns the size of the first group
inline int group_linearize_knots(const HostProblem& H, std::vector<int>& order) {
  order.clear();
  const int n = (int)H.knot_set.size();
  auto uses = [&](int t) {
    const EmpcCostSet& s = H.sets[H.knot_set[t]];
    return s.ncosts > 0 && (s.costs[0].reserved & 1);
  };
  for (int t = 0; t < n; ++t)
    if (!uses(t)) order.push_back(t);
  const int n_lean = (int)order.size();
  for (int t = 0; t < n; ++t)
    if (uses(t)) order.push_back(t);
  return n_lean;
}

inline void init_traj_state(TrajState& s, const EmpcSolverParams& prm, int maxiter, bool is_feasible_arg,
                            const TrajState* prev) {
  traj_state_init(s, prm, maxiter, is_feasible_arg, prev);
}
}  // namespace empc
