// empc_backward4.hpp -- HOT-B kernel body, matrix-core form with zero-padded LDS tiles (the shipped form).
//
// Mathematics and MFMA tiling (the r01 form it grew out of lives in tests/csrc/superseded/ as an emulator cross-check): (W = Vxx' [Fx Fu], Q = H + [Fx Fu]^T [W | Vx'],
// Vxx = Qxx - Qxu K on v_mfma_f64_16x16x4_f64; LLT of Quu, gain solves, symmetrisation, gap terms, regularisation retry:
// crocoddyl SolverDDP::backwardPass / computeGains; call sites src/sbfddp.cpp:244,256,332).  What changed is
// how the operands are addressed.  backward3 was bound by its instruction stream (~4k instructions per knot, a quarter of
// them arithmetic): every operand fetch clamped its indices and selected zeros at the tile edges, every accumulator store
// was predicated through a computed destination.  Here
//   * every matrix lives in LDS in a zero-padded tile-shaped array (V 32 x 21, Q 32 x 33, -K 12 x 33, ...): an operand
//     fetch is one ds_read at (per-lane base, fixed once) + immediate offset, with no select -- padding rows / columns
//     contribute exact zeros, and where a padded operand meets finite garbage of the other operand the product is zero;
//   * accumulator tiles are stored whole into the padded arrays (garbage lands in padding that nothing reads as data);
//   * W never goes through LDS: in the accumulator layout D[i = 4 r + lane / 16][j = lane % 16] register r of tile
//     (mt, nt) IS the B operand B[k = lane / 16][j] of k step 4 mt + r of the next product;
//   * the gains stay in the registers of the lane that solved them (column j of K): written to global memory, to LDS as
//     -K (B operand of the Vxx update) and used for Vx without a round trip.
#pragma once
#include "empc_kernels.hpp"
#include "empc_boxqp.hpp"

namespace empc {

#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define BWD_STAMP(i)                                              \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    bst[i] += now_ - bst[15];                                     \
    bst[15] = now_;                                               \
  } while (0)
#else
#define BWD_STAMP(i) \
  do {               \
  } while (0)
#endif

// Scheduling fence: nothing moves across it.  With 256 registers in use the compiler otherwise turns a short loop over LDS
// values into load, wait, use, load, wait, use ... -- one exposed LDS round trip (~100 cycles) per element; a block of
// loads, a fence, then the arithmetic in its original order costs one.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define BWD_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BWD_FENCE() \
  do {              \
  } while (0)
#endif

// A value pinned in program order: an empty volatile asm statement that reads and "writes" it.  `each`'s scheduling fences only bind
// the machine scheduler; the selection DAG orders pure arithmetic by register pressure and moves it across them (seen in the ISA:
// the reciprocal square roots of the Cholesky columns sank below the matrix-core instructions they were meant to run under).
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define BWD_PIN(x) asm volatile("" : "+v"(x))
#else
#define BWD_PIN(x) \
  do {             \
  } while (0)
#endif

// two doubles that travel together (one 16-byte global load, one 16-byte LDS write): records are 128-byte aligned
// (a native vector type on the device: an array of 16-byte structs indexed by an unrolled loop stays in scratch memory)
#if defined(__HIPCC__)
typedef double Bwd4Pair __attribute__((ext_vector_type(2)));
#else
struct alignas(16) Bwd4Pair {
  double a, b;
};
#endif

template <class DM>
struct Bwd4Smem {
  static constexpr int n = DM::NDX, m = DM::NU, nm = n + m;
  static constexpr int MTN = (n + 15) / 16, MTQ = (nm + 15) / 16, NTQ = (nm + 1 + 15) / 16;
  static constexpr int KSN = (n + 3) / 4, KSM = (m + 3) / 4;
  static constexpr int VS = 4 * KSN + 1;        // row stride of V (odd: the 16 rows of an operand fetch spread over the banks)
  static constexpr int QS = 16 * NTQ + 1;       // row stride of Q
  static constexpr int WS = 16 * MTN + 1;       // row stride of the unsymmetrised Vxx
  static constexpr int KS = 16 * MTN + 1;       // row stride of -K
  // Q rows: the A-operand fetch of the Vxx stage touches 16 MTN rows, the accumulator store rows < nm rounded up to 4
  static constexpr int QROWS = (16 * MTN > (nm + 3) / 4 * 4) ? 16 * MTN : (nm + 3) / 4 * 4;
  // EMPC_BWD_SYMTILES: a 16 x 16 tile (mt, nt) of Q / of Vxx whose every entry is the mirror image of an entry of a computed
  // tile and is read by nobody: below the diagonal tiles and entirely left of column n (lower-left of Qxx, or Qux)
  static constexpr bool tile_skipped(int mt, int nt) { return EMPC_BWD_SYMTILES && nt < mt && 16 * nt + 15 < n; }
  // entry (i, j) of the n x n value-function Hessian lies in a skipped tile
  static constexpr bool entry_skipped(int i, int j) { return tile_skipped(i / 16, j / 16); }
  // EMPC_BWD_MFMA4: row groups (four rows = one accumulator register of a tile) that hold data
  static constexpr int RGN = (n + 3) / 4, RGQ = (nm + 3) / 4;
  // ... and a row group of Q that nobody reads: rows of Qux / Quu (all >= n) in a column tile left of column n
  static constexpr bool group_skipped(int rg, int nt) { return tile_skipped(rg / 4, nt) || (EMPC_BWD_MFMA4 && 4 * rg >= n && 16 * nt + 15 < n); }
  static constexpr int OFF_REC = 0;                                   // the record, flat, in whole 64-double rows
  // Aliases inside the record area (LDS per wavefront decides how many trajectories a CU holds: 4 only below 40 KB, and the
  // 11-dof class was at 53 KB = 3 per CU = two rounds of workgroups for 1024 trajectories):
  //  * W, the unsymmetrised Vxx, lives from the end of the Vxx stage to the symmetrise stage; [Fx Fu] and H (the record
  //    in front of Lx / Lu / gap) are dead after the Q stage and the next record arrives after the symmetrise stage;
  //  * the inverse of the free block of Quu (box solvers) sits behind W, also in front of OFF_LX;
  //  * the prologue's partial sums are used before the first record is written.
  static constexpr int WROWS = (n + 3) / 4 * 4;                       // accumulator rows of the Vxx tiles that are stored
  static constexpr int OFF_W = OFF_REC;                               // [WROWS][WS]
  static constexpr int OFF_HINV = OFF_W + WROWS * WS;                 // m x m
  static constexpr int OFF_PRO = OFF_REC;                             // prologue reductions: 3 x 64
  static_assert(OFF_HINV + m * m <= DM::OFF_LX, "W and Hinv must fit in front of the part of the record that stays live");
#if EMPC_BWD_GLDS
  static constexpr int OFF_V = (DM::REC + 127) / 128 * 128 + 2;       // behind record buffer 0 and its zero word
#elif EMPC_BWD_R4B
  static constexpr int OFF_V = (DM::REC + 127) / 128 * 128;           // [16 MTN][VS], zero outside n x n
#else
  static constexpr int OFF_V = (DM::REC + 63) / 64 * 64;              // [16 MTN][VS], zero outside n x n
#endif
  static_assert(5 * 64 <= OFF_V, "prologue sums / end-of-pass sums inside the record area");
  static constexpr int OFF_VX = OFF_V + 16 * MTN * VS;                // [4 KSN], zero beyond n
  static constexpr int OFF_Q = OFF_VX + 4 * KSN;                      // [QROWS][QS]
  static constexpr int OFF_KN = OFF_Q + QROWS * QS;                   // [4 KSM][KS], zero outside m x n
  static constexpr int OFF_KF = OFF_KN + 4 * KSM * KS;                // k (m), Quuk (m)
  static constexpr int OFF_RED = OFF_KF + 2 * m;                      // 3 x 32 partial sums
  static constexpr int OFF_FLAG = OFF_RED + 96;
  static constexpr int OFF_ZERO = OFF_FLAG + 2;                       // a word that holds 0.0 (H entries outside the matrix)
  static constexpr int SIZE0 = (OFF_ZERO + 2 + 1) / 2 * 2;
  // EMPC_BWD_GLDS: two record buffers of RB doubles (whole 1-KiB pieces of the LDS-DMA) + a zero word each at [RB]; buffer 0 is the
  // record area above (OFF_V lies behind its zero word), buffer 1 follows everything else
  static constexpr int RB = (DM::REC + 127) / 128 * 128;
  static constexpr bool GLDS = EMPC_BWD_GLDS && 4 * sizeof(double) * (SIZE0 + RB + 2) <= 160 * 1024;
  static constexpr int OFF_REC2 = SIZE0;
  static constexpr int SIZE = GLDS ? SIZE0 + RB + 2 : SIZE0;
  static_assert(4 * sizeof(double) * SIZE <= 160 * 1024, "four trajectories (one per SIMD) must fit the LDS of a CU: a class beyond that runs in two rounds");
};

// BOX: the instantiation for crocoddyl's SolverBoxFDDP / SolverBoxDDP (box-QP gains); the squash-box solver's instantiation
// carries none of that code, so its register allocation is that of the plain pass
template <class DM, bool BOX, class Exec>
EMPC_HD void backward_traj4(Exec& ex, const DevBuffers& D, int b, double* smem) {
  typedef Bwd4Smem<DM> SM;
  constexpr int NL = 64;
  constexpr int n = DM::NDX, m = DM::NU, nm = n + m, REC = DM::REC;
  static_assert(nm <= 47 && n <= 32, "tile counts of the matrix-core backward pass");
#if EMPC_BWD_R4B
  constexpr int PRE = (REC + 2 * NL - 1) / (2 * NL);  // prefetch register PAIRS per lane: the record moves in 16-byte pieces
#else
  constexpr int PRE = (REC + NL - 1) / NL;  // prefetch registers per lane
#endif
  constexpr int MTN = SM::MTN, MTQ = SM::MTQ, NTQ = SM::NTQ, KSN = SM::KSN, KSM = SM::KSM;
  constexpr int VS = SM::VS, QS = SM::QS, WS = SM::WS, KS = SM::KS;
  TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const int T = D.T;
  double* rec = smem + SM::OFF_REC;
  double* V = smem + SM::OFF_V;
  double* vx = smem + SM::OFF_VX;
  double* Q = smem + SM::OFF_Q;
  double* W = smem + SM::OFF_W;
  double* Kn = smem + SM::OFF_KN;
  double* kf = smem + SM::OFF_KF;
  double* red = smem + SM::OFF_RED;
  double* flag = smem + SM::OFF_FLAG;
  double* pro = smem + SM::OFF_PRO;
  double* Hinv = smem + SM::OFF_HINV;
  const double* tape = D.tape + (size_t)b * (T + 1) * REC;

  // ---- prologue: cost, gap norms, feasibility (as backward3) ---------------------------------------------------------
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
      pro[64 + lane] = mx;
      pro[128 + lane] = l1;
    });
    ex.sync();
    double tot_c = 0, tot_mx = 0, tot_l1 = 0;
    for (int i = 0; i < NL; ++i) {
      tot_c += pro[i];
      tot_mx = fmax(tot_mx, pro[64 + i]);
      tot_l1 += pro[128 + i];
    }
    ex.sync();
    cost = tot_c;
    if (!is_feasible) is_feasible = (tot_mx < D.gaptol) ? 1 : 0;
    gapnorm = (P.prm.gap_norm == EMPC_GAP_L1) ? tot_l1 : tot_mx;
  }
  const bool infeas = !is_feasible;

  // ---- loop-invariant addressing -----------------------------------------------------------------------------------------
  // tile coordinates of this lane: li = row (A operand) / column (B operand, accumulator), lq = k offset / accumulator row group
  // H entries behind the accumulators of the Q stage: offsets into the record, or the zero word outside H
  int hidx[Exec::SLOTS][MTQ][NTQ][4];
  ex.each([&](int lane, int sl) {
    const int lj = lane % 16, lq = lane / 16;
#pragma unroll
    for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * mt + 4 * r + lq, j = 16 * nt + lj;
          int idx = SM::GLDS ? SM::RB : SM::OFF_ZERO;  // (GLDS: offsets relative to the current record buffer, zero word behind it)
          if (i < n && j < nm)
#if EMPC_REC_TRI
            idx = (j < n) ? DM::lxx(i, j) : DM::lxu(i, j - n);
#else
            idx = DM::OFF_HX + i * nm + j;
#endif
          else if (i < n && j == nm)
            idx = DM::OFF_LX + i;
          else if (i >= n && i < nm && j >= n && j < nm)
#if EMPC_REC_TRI
            idx = DM::luu(i - n, j - n);
#else
            idx = DM::OFF_LUU + (i - n) * m + (j - n);
#endif
          else if (i >= n && i < nm && j == nm)
            idx = DM::OFF_LU + (i - n);
          hidx[sl][mt][nt][r] = idx;
        }
    // zero padding, written once: everything but the record area
    for (int i = lane; i < SM::SIZE - SM::OFF_V; i += NL) smem[SM::OFF_V + i] = 0.0;
    if constexpr (SM::GLDS)
      if (lane == 0) smem[SM::RB] = 0.0;  // zero word of record buffer 0 (the one of buffer 1 lies in the range above)
  });
  ex.sync();
#if EMPC_BWD_R4B
  // symmetrise stage: entry (i, j), i <= j, of the upper triangle per lane and round; offsets of W[i][j], W[j][i], V[i][j], V[j][i]
  // (-1: no entry).  One read pair and one average serve both halves (a + b == b + a bit for bit).
  constexpr int NTRI = n * (n + 1) / 2, NSY = (NTRI + NL - 1) / NL;
  int sy_w[Exec::SLOTS][NSY][2], sy_v[Exec::SLOTS][NSY][2];
  ex.each([&](int lane, int sl) {
#pragma unroll
    for (int q = 0; q < NSY; ++q) {
      int idx = lane + q * NL, i = 0;
      const bool on = idx < NTRI;
      if (!on) idx = 0;
      while (idx >= n - i) {
        idx -= n - i;
        ++i;
      }
      const int j = i + idx;
      sy_w[sl][q][0] = on ? i * WS + j : -1;
      sy_w[sl][q][1] = SM::entry_skipped(j, i) ? i * WS + j : j * WS + i;  // (i <= j: (i, j) itself is never in a skipped tile)
      sy_v[sl][q][0] = i * VS + j;
      sy_v[sl][q][1] = (i == j) ? -1 : j * VS + i;  // -1 marks a diagonal entry: regularised, written once
    }
  });
#endif

  double xreg = st.xreg, ureg = st.ureg;
  double dg_u = 0, dq_u = 0, dg_f = 0, dq_f = 0, qu2 = 0;
  // The sums of the expected improvement and of the stopping criterion (sum_t Qu.k, k.Quu k, Qu.Qu, Vx.f, f.Vxx f) are kept
  // per lane -- lane i adds ITS term of every knot -- and the lanes' totals are added once, after the last knot.  (Rounds 2-3:
  // every lane read all 2 x (m + n) terms of a knot back from LDS and added them, 1.2k of a knot's 13.5k cycles.)
  double dgu_l[Exec::SLOTS], dqu_l[Exec::SLOTS], qu2_l[Exec::SLOTS], dgf_l[Exec::SLOTS], dqf_l[Exec::SLOTS];
  bool failed_final = false;
  while (true) {
    bool fail = false;
    dg_u = dq_u = dg_f = dq_f = qu2 = 0;
    ex.each([&](int lane, int sl) { dgu_l[sl] = dqu_l[sl] = qu2_l[sl] = dgf_l[sl] = dqf_l[sl] = 0.0; });
    // ---- terminal node ---------------------------------------------------------------------------------------
    {
      const double* r = tape + (size_t)T * REC;
      ex.each([&](int lane, int sl) {
        for (int i = lane; i < n * n; i += NL)
#if EMPC_REC_TRI
          V[(i / n) * VS + (i % n)] = r[DM::lxx(i / n, i % n)] + (((i / n) == (i % n)) ? xreg : 0.0);
#else
          V[(i / n) * VS + (i % n)] = r[DM::OFF_LXX + (i / n) * nm + (i % n)] + (((i / n) == (i % n)) ? xreg : 0.0);
#endif
        if (lane < n) {
          vx[lane] = r[DM::OFF_LX + lane];
          red[64 + lane] = r[DM::OFF_GAP + lane];  // gap of node T
        }
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane >= n) return;
        double a_ = 0;
        if (infeas)
          for (int j = 0; j < n; ++j) a_ += V[lane * VS + j] * red[64 + j];
        const double nv = vx[lane] + (infeas ? a_ : 0.0);
        D.Vf[((size_t)b * (T + 1) + T) * n + lane] = a_;
        D.Vx[((size_t)b * (T + 1) + T) * n + lane] = nv;
        if (infeas) {
          dgf_l[sl] -= nv * red[64 + lane];
          dqf_l[sl] += red[64 + lane] * a_;
        }
        Q[lane] = nv;              // staged: vx is still being read by the other lanes
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane < n) vx[lane] = Q[lane];
      });
      ex.sync();
    }
    // first record of the sweep
    if constexpr (SM::GLDS) {
      ex.async_wait();  // (a pass that failed left the copy of a record it never used in flight)
      ex.template async_rows<SM::RB / 128>(smem + SM::OFF_REC, tape + (size_t)(T - 1) * REC);  // into buffer 0
    }
#if EMPC_BWD_R4B
    Bwd4Pair pre[Exec::SLOTS][PRE];
    ex.each([&](int lane, int sl) {
      const Bwd4Pair* r = reinterpret_cast<const Bwd4Pair*>(tape + (size_t)(T - 1) * REC);
#pragma unroll
      for (int q = 0; q < PRE; ++q) pre[sl][q] = r[lane + q * NL];  // whole 128-double rows: they end inside the next record
    });
#else
    double pre[Exec::SLOTS][PRE];
    ex.each([&](int lane, int sl) {
      const double* r = tape + (size_t)(T - 1) * REC;
#pragma unroll
      for (int q = 0; q < PRE; ++q) pre[sl][q] = r[lane + q * NL];  // whole rows: the tape has one row of slack
    });
#endif
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long bst[16];
    for (int i = 0; i < 16; ++i) bst[i] = 0;
    bst[15] = __builtin_readcyclecounter();
#endif
    // Outputs of a knot (gains, Vx, Vxx f) stay in registers and go to memory at the top of the NEXT knot, right before the
    // record prefetch is issued: loads and stores share one in-order counter, so a store issued late in a knot would be
    // waited for when the prefetched record is consumed (measured: ~2k cycles per knot).
    double Kc[Exec::SLOTS][m], vfo[Exec::SLOTS], vxo[Exec::SLOTS];
#if EMPC_BWD_FUSE
    bool pdl[Exec::SLOTS];  // the LLT's verdict as every lane found it (the factorisation is redundant across the lanes)
    ex.each([&](int lane, int sl) { pdl[sl] = true; });
#endif
#if EMPC_BWD_VPTR
    // this lane's slots of knot T - 1 (the first knot flushed); every flush steps them back one knot.  Pinned: opaque per-lane values
    // stay in vector registers and their increments are vector adds.
    // (global address space stated: a pointer that went through the pin is generic to the compiler, and a FLAT store counts on
    //  the LDS counter as well -- every later wait for an LDS read would wait for it)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(1))) double* GPtr;
#else
    typedef double* GPtr;
#endif
    GPtr pK[Exec::SLOTS], pVf[Exec::SLOTS], pVx[Exec::SLOTS];
    ex.each([&](int lane, int sl) {
      pK[sl] = (GPtr)((lane == n) ? D.kff + ((size_t)b * T + (T - 1)) * m : D.K + ((size_t)b * T + (T - 1)) * m * n + (lane < n ? lane : 0));
      pVf[sl] = (GPtr)(D.Vf + ((size_t)b * (T + 1) + (T - 1)) * n + (lane < n ? lane : 0));
      pVx[sl] = (GPtr)(D.Vx + ((size_t)b * (T + 1) + (T - 1)) * n + (lane < n ? lane : 0));
      BWD_PIN(pK[sl]);
      BWD_PIN(pVf[sl]);
      BWD_PIN(pVx[sl]);
    });
    auto flush_outputs = [&](int tk, int lane, int sl) {  // (tk: the knot the pointers stand at -- T - 1, T - 2, ... in this order)
      if (lane < n) {
#pragma unroll
        for (int i = 0; i < m; ++i) pK[sl][i * n] = Kc[sl][i];
        *pVf[sl] = vfo[sl];
        *pVx[sl] = vxo[sl];
      } else if (lane == n) {
#pragma unroll
        for (int i = 0; i < m; ++i) pK[sl][i] = Kc[sl][i];
      }
      pK[sl] -= (lane == n) ? m : m * n;
      pVf[sl] -= n;
      pVx[sl] -= n;
    };
#else
    auto flush_outputs = [&](int tk, int lane, int sl) {
      if (lane < n) {
        double* Kg = D.K + ((size_t)b * T + tk) * m * n;
#pragma unroll
        for (int i = 0; i < m; ++i) Kg[i * n + lane] = Kc[sl][i];
        D.Vf[((size_t)b * (T + 1) + tk) * n + lane] = vfo[sl];
        D.Vx[((size_t)b * (T + 1) + tk) * n + lane] = vxo[sl];
      } else if (lane == n) {
#pragma unroll
        for (int i = 0; i < m; ++i) D.kff[((size_t)b * T + tk) * m + i] = Kc[sl][i];
      }
    };
#endif
    // EMPC_BWD_GLDS: the knot body exists twice, once per record buffer (a lambda over the buffer index, called alternately), so
    // that every LDS address of a knot is base + immediate as in the default build; `BWD_KNOT_EXIT` leaves the knot loop
#if EMPC_BWD_GLDS
#define BWD_KNOT_EXIT return false
    auto knot = [&](const int t, auto RBUF) -> bool {
      constexpr int rbuf = decltype(RBUF)::value;  // the buffer that holds this knot's record; the next record goes to the other
      double* const rec = smem + ((SM::GLDS && rbuf) ? SM::OFF_REC2 : SM::OFF_REC);  // (a class without the second buffer stages through registers)
      double* const W = rec + SM::OFF_W;
      double* const Hinv = rec + SM::OFF_HINV;
#else
#define BWD_KNOT_EXIT break
    for (int t = T - 1; t >= 0; --t) {
#endif
      BWD_STAMP(7);
      if constexpr (SM::GLDS) {
        // the record of this knot was requested a whole knot ago, straight into buffer `rbuf`; the other buffer (the previous
        // knot's record, W, Hinv: all dead) takes the next one.  Stores first: they share the counter the wait reads.
        ex.async_wait();
        ex.each([&](int lane, int sl) {
          if (t < T - 1) flush_outputs(t + 1, lane, sl);
        });
#if EMPC_BWD_GLDS
        if (t > 0) ex.template async_rows<SM::RB / 128>(smem + (rbuf ? SM::OFF_REC : SM::OFF_REC2), tape + (size_t)(t - 1) * REC);
#endif
      } else
      ex.each([&](int lane, int sl) {
#if EMPC_BWD_R4B
#pragma unroll
        for (int q = 0; q < PRE; ++q) reinterpret_cast<Bwd4Pair*>(rec)[lane + q * NL] = pre[sl][q];
        if (t < T - 1) flush_outputs(t + 1, lane, sl);
        if (t > 0) {
          const Bwd4Pair* r = reinterpret_cast<const Bwd4Pair*>(tape + (size_t)(t - 1) * REC);
#pragma unroll
          for (int q = 0; q < PRE; ++q) pre[sl][q] = r[lane + q * NL];
        }
#else
#pragma unroll
        for (int q = 0; q < PRE; ++q) rec[lane + q * NL] = pre[sl][q];
        if (t < T - 1) flush_outputs(t + 1, lane, sl);
        if (t > 0) {
          const double* r = tape + (size_t)(t - 1) * REC;
#pragma unroll
          for (int q = 0; q < PRE; ++q) pre[sl][q] = r[lane + q * NL];
        }
#endif
      });
      ex.sync();
      BWD_STAMP(0);
      // accumulators of the Q stage start from H (independent of the value function: issued before the W product)
      double accQ[Exec::SLOTS][MTQ][NTQ][4];
      ex.each([&](int lane, int sl) {
#pragma unroll
        for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!SM::group_skipped(4 * mt + r, nt) && (!EMPC_BWD_MFMA4 || 4 * mt + r < SM::RGQ))
                accQ[sl][mt][nt][r] = (SM::GLDS ? rec : smem)[hidx[sl][mt][nt][r]];
      });
      // W = V' A, A = [Fx Fu] (flat in the record, row stride nm).  k rows >= n meet the zero columns of V; columns >= nm are
      // finite garbage that ends in columns nobody uses (column nm is replaced by Vx' below).
      double accW[Exec::SLOTS][MTN][NTQ][4];
      ex.each([&](int lane, int sl) {
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) accW[sl][mt][nt][r] = 0.0;
      });
#if EMPC_BWD_OVERLAP
      // EMPC_BWD_OVERLAP.  NTX column tiles lie wholly left of column n ("x tiles": Qxx | Qux); the others ("u tiles") hold Qxu | Quu |
      // Qx, Qu.  Phase 1: W and Q restricted to the u tiles -> LDS.  Phase 2: the x tiles, PER_PIECE instructions after each of the
      // 3 m pieces of computeGains (m Cholesky columns, m rows forward, m rows backward); what is left of either list runs at the end.
      // A operands per k step: one per 16-row tile (16 x 16 x 4 form) or one per 4-row group (EMPC_BWD_MFMA4: the group's rows in
      // every block, instructions of 16 instead of 64 cycles -- two per piece).
      constexpr int NTX = n / 16;
#if EMPC_BWD_MFMA4
      constexpr int AW = SM::RGN, AQ = SM::RGQ, PER_PIECE = 2;
#else
      constexpr int AW = MTN, AQ = MTQ, PER_PIECE = 1;
#endif
      if constexpr (!BOX && NTX > 0) {
        auto a_row = [](int a_, int li) { return EMPC_BWD_MFMA4 ? 4 * a_ + li % 4 : 16 * a_ + li; };  // row of A operand a_ held by lane column li
        auto skipped = [](int a_, int nt) { return EMPC_BWD_MFMA4 ? SM::group_skipped(a_, nt) : SM::tile_skipped(a_, nt); };
        auto product = [&](auto& aop, int a_, auto& bop, int nt, auto& acc) {
#if EMPC_BWD_MFMA4
          ex.mfma4(aop, a_, bop, nt, acc, a_ / 4, nt, a_ % 4);
#else
          ex.mfma(aop, a_, bop, nt, acc, a_, nt);
#endif
        };
        auto loadW = [&](double (&aop)[2][Exec::SLOTS][AW], double (&bop)[2][Exec::SLOTS][NTQ], int ks, int buf, int nt0, int nt1) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int a_ = 0; a_ < AW; ++a_) aop[buf][sl][a_] = V[a_row(a_, li) * VS + 4 * ks + lq];
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt)
              if (nt >= nt0 && nt < nt1) bop[buf][sl][nt] = rec[DM::OFF_A + (4 * ks + lq) * nm + 16 * nt + li];
          });
        };
        auto loadQ = [&](double (&aop)[2][Exec::SLOTS][AQ], double (&bop)[2][Exec::SLOTS][NTQ], int ks, int buf, int nt0, int nt1) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int a_ = 0; a_ < AQ; ++a_) aop[buf][sl][a_] = rec[DM::OFF_A + (4 * ks + lq) * nm + a_row(a_, li)];
            const double vxk = vx[4 * ks + lq];
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt)
              if (nt >= nt0 && nt < nt1) bop[buf][sl][nt] = (16 * nt + li == nm) ? vxk : accW[sl][ks / 4][nt][ks % 4];
          });
        };
        // ---- phase 1: u tiles ----------------------------------------------------------------------------------------
        {
          double aopW[2][Exec::SLOTS][AW], bopW[2][Exec::SLOTS][NTQ];
          loadW(aopW, bopW, 0, 0, NTX, NTQ);
#pragma unroll
          for (int ks = 0; ks < KSN; ++ks) {
            if (ks + 1 < KSN) loadW(aopW, bopW, ks + 1, (ks + 1) & 1, NTX, NTQ);
#pragma unroll
            for (int a_ = 0; a_ < AW; ++a_)
#pragma unroll
              for (int nt = NTX; nt < NTQ; ++nt) product(aopW[ks & 1], a_, bopW[ks & 1], nt, accW);
          }
        }
        {
          double aopQ[2][Exec::SLOTS][AQ], bopQ[2][Exec::SLOTS][NTQ];
          loadQ(aopQ, bopQ, 0, 0, NTX, NTQ);
#pragma unroll
          for (int ks = 0; ks < KSN; ++ks) {
            if (ks + 1 < KSN) loadQ(aopQ, bopQ, ks + 1, (ks + 1) & 1, NTX, NTQ);
#pragma unroll
            for (int a_ = 0; a_ < AQ; ++a_)
#pragma unroll
              for (int nt = NTX; nt < NTQ; ++nt)
                if (!skipped(a_, nt)) product(aopQ[ks & 1], a_, bopQ[ks & 1], nt, accQ);
          }
        }
        ex.each([&](int lane, int sl) {
          const int lj = lane % 16, lq = lane / 16;
#pragma unroll
          for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
            for (int nt = NTX; nt < NTQ; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (16 * mt + 4 * r < nm) Q[(16 * mt + 4 * r + lq) * QS + 16 * nt + lj] = accQ[sl][mt][nt][r];
        });
        ex.sync();
        BWD_STAMP(2);
        // ---- phase 2: the x tiles between the pieces of computeGains ------------------------------------------------------
        double Lq[Exec::SLOTS][m * (m + 1) / 2], rhs[Exec::SLOTS][m];
        bool pd[Exec::SLOTS];
        ex.each([&](int lane, int sl) {
#pragma unroll
          for (int i = 0; i < m; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Lq[sl][i * (i + 1) / 2 + j] = Q[(n + i) * QS + n + j];
#pragma unroll
          for (int i = 0; i < m; ++i) rhs[sl][i] = Q[((lane < n) ? lane : (n + i)) * QS + ((lane < n) ? (n + i) : nm)];
          BWD_FENCE();
#pragma unroll
          for (int i = 0; i < m; ++i) Lq[sl][i * (i + 1) / 2 + i] += ureg;
          pd[sl] = true;
        });
        // piece c of computeGains: chol_packed's column c | chol_solve_packed's forward row c - m | its backward row 3 m - 1 - c
        // (empc_dev_model.hpp: the same operations in the same order on every entry)
        auto piece = [&](int c) {
          ex.each([&](int lane, int sl) {
            double* L = Lq[sl];
            double* b_ = rhs[sl];
            if (c < m) {
              const int j = c;
              double s_ = L[j * (j + 1) / 2 + j];
#pragma unroll
              for (int k = 0; k < m; ++k)
                if (k < j) s_ -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
              if (!(s_ > 0.0) || is_nan(s_)) pd[sl] = false;
              const double inv = frsqrt(s_);
              L[j * (j + 1) / 2 + j] = inv;
#pragma unroll
              for (int i = 0; i < m; ++i)
                if (i > j) {
                  double t_ = L[i * (i + 1) / 2 + j];
#pragma unroll
                  for (int k = 0; k < m; ++k)
                    if (k < j) t_ -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                  L[i * (i + 1) / 2 + j] = t_ * inv;
                  BWD_PIN(L[i * (i + 1) / 2 + j]);
                }
              BWD_PIN(L[j * (j + 1) / 2 + j]);
              if (c == m - 1 && lane == 0) flag[0] = pd[sl] ? 0.0 : 1.0;
#if EMPC_BWD_FUSE
              if (c == m - 1) pdl[sl] = pd[sl];
#endif
            } else if (c < 2 * m) {
              const int i = c - m;
              if (lane <= n) {
                double s_ = b_[i];
#pragma unroll
                for (int k = 0; k < m; ++k)
                  if (k < i) s_ -= L[i * (i + 1) / 2 + k] * b_[k];
                b_[i] = s_ * L[i * (i + 1) / 2 + i];
                BWD_PIN(b_[i]);
              }
            } else {
              const int i = 3 * m - 1 - c;
              if (lane <= n) {
                double s_ = b_[i];
#pragma unroll
                for (int k = 0; k < m; ++k)
                  if (k > i) s_ -= L[k * (k + 1) / 2 + i] * b_[k];
                b_[i] = s_ * L[i * (i + 1) / 2 + i];
                BWD_PIN(b_[i]);
              }
            }
          });
        };
        int pc = 0;  // next piece (a compile-time constant at every use once the loops below are unrolled)
        int issued = 0;
        {
          double aopW[2][Exec::SLOTS][AW], bopW[2][Exec::SLOTS][NTQ];
          loadW(aopW, bopW, 0, 0, 0, NTX);
#pragma unroll
          for (int ks = 0; ks < KSN; ++ks) {
            if (ks + 1 < KSN) loadW(aopW, bopW, ks + 1, (ks + 1) & 1, 0, NTX);
#pragma unroll
            for (int a_ = 0; a_ < AW; ++a_)
#pragma unroll
              for (int nt = 0; nt < NTX; ++nt) {
                product(aopW[ks & 1], a_, bopW[ks & 1], nt, accW);
                if (++issued % PER_PIECE == 0 && pc < 3 * m) piece(pc++);
              }
          }
        }
        {
          double aopQ[2][Exec::SLOTS][AQ], bopQ[2][Exec::SLOTS][NTQ];
          loadQ(aopQ, bopQ, 0, 0, 0, NTX);
#pragma unroll
          for (int ks = 0; ks < KSN; ++ks) {
            if (ks + 1 < KSN) loadQ(aopQ, bopQ, ks + 1, (ks + 1) & 1, 0, NTX);
#pragma unroll
            for (int a_ = 0; a_ < AQ; ++a_)
#pragma unroll
              for (int nt = 0; nt < NTX; ++nt)
                if (!skipped(a_, nt)) {
                  product(aopQ[ks & 1], a_, bopQ[ks & 1], nt, accQ);
                  if (++issued % PER_PIECE == 0 && pc < 3 * m) piece(pc++);
                }
          }
        }
#pragma unroll
        for (int c = 0; c < 3 * m; ++c)
          if (c >= pc) piece(c);
        ex.each([&](int lane, int sl) {
          if (lane <= n) {
#pragma unroll
            for (int i = 0; i < m; ++i) Kc[sl][i] = rhs[sl][i];
            if (lane < n) {
#pragma unroll
              for (int i = 0; i < m; ++i) Kn[i * KS + lane] = -rhs[sl][i];
            } else {
#pragma unroll
              for (int i = 0; i < m; ++i) kf[i] = rhs[sl][i];
            }
          }
        });
      } else
#endif
      {
      // (operands of k step ks + 1 are requested from LDS before the matrix-core instructions of step ks are issued: the
      //  stage fences of `each` would otherwise expose one LDS round trip per k step)
#if EMPC_BWD_MFMA4
      {
        // one instruction per (row group, column tile, k step): A = the group's four rows of V' in every block
        double aopW[2][Exec::SLOTS][SM::RGN], bopW[2][Exec::SLOTS][NTQ];
        auto loadW = [&](int ks, int buf) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int rg = 0; rg < SM::RGN; ++rg) aopW[buf][sl][rg] = V[(4 * rg + li % 4) * VS + 4 * ks + lq];
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) bopW[buf][sl][nt] = rec[DM::OFF_A + (4 * ks + lq) * nm + 16 * nt + li];
          });
        };
        loadW(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
          if (ks + 1 < KSN) loadW(ks + 1, (ks + 1) & 1);
#pragma unroll
          for (int rg = 0; rg < SM::RGN; ++rg)
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) ex.mfma4(aopW[ks & 1], rg, bopW[ks & 1], nt, accW, rg / 4, nt, rg % 4);
        }
      }
#else
      {
        double aopW[2][Exec::SLOTS][MTN], bopW[2][Exec::SLOTS][NTQ];
        auto loadW = [&](int ks, int buf) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) aopW[buf][sl][mt] = V[(16 * mt + li) * VS + 4 * ks + lq];
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) bopW[buf][sl][nt] = rec[DM::OFF_A + (4 * ks + lq) * nm + 16 * nt + li];
          });
        };
        loadW(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
          if (ks + 1 < KSN) loadW(ks + 1, (ks + 1) & 1);
#pragma unroll
          for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) ex.mfma(aopW[ks & 1], mt, bopW[ks & 1], nt, accW, mt, nt);
        }
      }
#endif
      BWD_STAMP(1);
      // Q = H + A^T [W | Vx'].  A^T[i][k] = A[k][i]: same address pattern as the B operand above.  B operand = the W
      // accumulators themselves (register r of tile (mt, nt) holds row 4 mt' + r ... of k step ks = 4 mt + r), with column
      // nm taken from Vx' (zero beyond n).  Rows >= n of W are exact zeros (zero rows of V), so the garbage A^T values of
      // k >= n contribute nothing.
#if EMPC_BWD_MFMA4
      {
        double aopQ[2][Exec::SLOTS][SM::RGQ], bopQ[2][Exec::SLOTS][NTQ];
        auto loadQ = [&](int ks, int buf) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int rg = 0; rg < SM::RGQ; ++rg) aopQ[buf][sl][rg] = rec[DM::OFF_A + (4 * ks + lq) * nm + 4 * rg + li % 4];
            const double vxk = vx[4 * ks + lq];
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) bopQ[buf][sl][nt] = (16 * nt + li == nm) ? vxk : accW[sl][ks / 4][nt][ks % 4];
          });
        };
        loadQ(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
          if (ks + 1 < KSN) loadQ(ks + 1, (ks + 1) & 1);
#pragma unroll
          for (int rg = 0; rg < SM::RGQ; ++rg)
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt)
              if (!SM::group_skipped(rg, nt)) ex.mfma4(aopQ[ks & 1], rg, bopQ[ks & 1], nt, accQ, rg / 4, nt, rg % 4);
        }
      }
#else
      {
        double aopQ[2][Exec::SLOTS][MTQ], bopQ[2][Exec::SLOTS][NTQ];
        auto loadQ = [&](int ks, int buf) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int mt = 0; mt < MTQ; ++mt) aopQ[buf][sl][mt] = rec[DM::OFF_A + (4 * ks + lq) * nm + 16 * mt + li];
            const double vxk = vx[4 * ks + lq];
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) bopQ[buf][sl][nt] = (16 * nt + li == nm) ? vxk : accW[sl][ks / 4][nt][ks % 4];
          });
        };
        loadQ(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
          if (ks + 1 < KSN) loadQ(ks + 1, (ks + 1) & 1);
#pragma unroll
          for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt)
              if (!SM::tile_skipped(mt, nt)) ex.mfma(aopQ[ks & 1], mt, bopQ[ks & 1], nt, accQ, mt, nt);
        }
      }
#endif
      // whole tiles into the padded Q array (column nm = Qx | Qu)
      ex.each([&](int lane, int sl) {
        const int lj = lane % 16, lq = lane / 16;
#pragma unroll
        for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              // only what is read back from LDS: rows < nm of the column tiles that reach column n (Qxu | Quu | Qx, Qu);
              // the Qxx block stays in the accumulators for the Vxx stage
              if (16 * nt + 15 >= n && 16 * mt + 4 * r < nm) Q[(16 * mt + 4 * r + lq) * QS + 16 * nt + lj] = accQ[sl][mt][nt][r];
      });
      ex.sync();
      BWD_STAMP(2);
      // computeGains: LLT(Quu + ureg I) in every lane; lane j < n solves column j of K = Quu^-1 Qxu^T, lane n solves k and
      // forms Quu k.  The columns stay in registers.
      // SolverBoxDDP::computeGains / SolverBoxFDDP::computeGains once the trajectory is feasible (both fall back to the plain
      // gains while the gaps are open: `!has_control_limits || !is_feasible_` -> SolverDDP::computeGains): k from the box QP over
      // [u_lb - us, u_ub - us] (one lane), K = (free block of Quu)^-1 Qux on the free controls and zero on the clamped ones, Qu
      // zeroed on the clamped ones (it enters the stopping criterion, the expected improvement and Vx below)
      const bool box_gains = BOX && is_feasible && (P.prm.solver_type == EMPC_SOLVER_BOXDDP || P.prm.solver_type == EMPC_SOLVER_BOXFDDP);
      if (box_gains) {
        ex.each([&](int lane, int sl) {
          if (lane != n) return;
          // H of the QP and the inverse of its free block live in LDS (W is dead between the symmetrise stage of the previous
          // knot and the Vxx stage of this one): 2 m^2 doubles in the registers of one lane put the whole kernel into scratch
          double* Hq = W;
#if EMPC_BOX_LDS
          // the QP's vectors behind Hinv, still in front of the live tail of the record (Lx, Lu, gap)
          static_assert(SM::OFF_HINV + m * m + 9 * m <= DM::OFF_LX, "box-QP working set inside the dead part of the record area");
          double* const qq = Hinv + m * m;
          double* const lbq = qq + m;
          double* const ubq = qq + 2 * m;
          double* const xq = qq + 3 * m;
          int* const fm = reinterpret_cast<int*>(qq + 4 * m);
          double* const ws = qq + 5 * m;  // g, xnew, dx, held free set
#else
          double qq[m], lbq[m], ubq[m], xq[m];
          int fm[m];
#endif
          const double* usg = D.us + ((size_t)b * T + t) * m;
          const double* kprev = D.kff + ((size_t)b * T + t) * m;  // k_[t] of the previous iteration: the QP's warm start
#pragma unroll
          for (int i = 0; i < m; ++i) {
#pragma unroll
            for (int j = 0; j < m; ++j) Hq[i * m + j] = Q[(n + (i > j ? i : j)) * QS + n + (i > j ? j : i)] + ((i == j) ? ureg : 0.0);
            qq[i] = Q[(n + i) * QS + nm];
            lbq[i] = P.u_lb[i] - usg[i];
            ubq[i] = P.u_ub[i] - usg[i];
            xq[i] = kprev[i];
          }
          const bool okq = box_qp_lane<m>(Hq, qq, lbq, ubq, xq, fm, Hinv, P.prm.boxqp_maxiter, P.prm.boxqp_th_acceptstep,
                                          P.prm.boxqp_th_grad, P.prm.boxqp_reg
#if EMPC_BOX_LDS
                                          , ws
#endif
          );
          flag[0] = okq ? 0.0 : 1.0;
#pragma unroll
          for (int i = 0; i < m; ++i) {
            kf[i] = -xq[i];
            Kc[sl][i] = -xq[i];
            if (!fm[i]) Q[(n + i) * QS + nm] = 0.0;
          }
        });
        ex.sync();
        ex.each([&](int lane, int sl) {
          if (lane >= n) return;
#pragma unroll
          for (int i = 0; i < m; ++i) {
            double a_ = 0;
#pragma unroll
            for (int l = 0; l < m; ++l) a_ += Hinv[i * m + l] * Q[lane * QS + n + l];
            Kc[sl][i] = a_;
            Kn[i * KS + lane] = -a_;
          }
        });
      } else
      ex.each([&](int lane, int sl) {
        // Quu and the lane's right-hand side: every LDS read issued before the arithmetic starts (BWD_FENCE); the
        // regularisation touches the diagonal only (an `+ 0.0` on the other entries is a real FP64 instruction)
        double Lq[m * (m + 1) / 2], rhs[m];
#pragma unroll
        for (int i = 0; i < m; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) Lq[i * (i + 1) / 2 + j] = Q[(n + i) * QS + n + j];
#pragma unroll
        for (int i = 0; i < m; ++i) rhs[i] = Q[((lane < n) ? lane : (n + i)) * QS + ((lane < n) ? (n + i) : nm)];
        BWD_FENCE();
#pragma unroll
        for (int i = 0; i < m; ++i) Lq[i * (i + 1) / 2 + i] += ureg;
        const bool pd = chol_packed<m>(Lq);
#if EMPC_BWD_FUSE
        pdl[sl] = pd;
#endif
        if (lane == 0) flag[0] = pd ? 0.0 : 1.0;
        if (lane <= n) {
          chol_solve_packed<m>(Lq, rhs);
#pragma unroll
          for (int i = 0; i < m; ++i) Kc[sl][i] = rhs[i];
          if (lane < n) {
#pragma unroll
            for (int i = 0; i < m; ++i) Kn[i * KS + lane] = -rhs[i];
          } else {
#pragma unroll
            for (int i = 0; i < m; ++i) kf[i] = rhs[i];
          }
        }
      });
      }
#if EMPC_BWD_FUSE
      const bool box_gains_f = BOX && is_feasible && (P.prm.solver_type == EMPC_SOLVER_BOXDDP || P.prm.solver_type == EMPC_SOLVER_BOXFDDP);
      if (!box_gains_f) {
        BWD_STAMP(8);
        if (!ex.first(pdl)) {  // (identical in every lane: the first one speaks for all)
          fail = true;
          BWD_KNOT_EXIT;
        }
        // k: lane n's solved column, to every lane as scalars
        double kk[m];
#pragma unroll
        for (int j = 0; j < m; ++j) kk[j] = ex.bcast(Kc, j, n);
        BWD_STAMP(9);
        // Quu k, one row per lane (same order of summation as the single-lane form)
        double qkl[Exec::SLOTS];
        ex.each([&](int lane, int sl) {
          qkl[sl] = 0.0;
          if (lane < m) {
            double qrow[m];
#pragma unroll
            for (int j = 0; j < m; ++j) qrow[j] = Q[(n + (lane > j ? lane : j)) * QS + n + (lane > j ? j : lane)];
            const double ql = Q[(n + lane) * QS + nm];
            double kl = kk[0];  // k[lane]: a select per control (pinned: left alone the compiler builds a table in scratch memory)
#pragma unroll
            for (int j = 1; j < m; ++j) {
              if (lane == j) kl = kk[j];
              BWD_PIN(kl);
            }
            BWD_FENCE();
            double a_ = 0;
#pragma unroll
            for (int j = 0; j < m; ++j) a_ += qrow[j] * kk[j];
            const double qk = a_ + ureg * kl;
            qkl[sl] = qk;
            dgu_l[sl] += ql * kl;
            dqu_l[sl] -= kl * qk;
            qu2_l[sl] += ql * ql;
          }
        });
        double qks[m];
#pragma unroll
        for (int l = 0; l < m; ++l) qks[l] = ex.bcast1(qkl, l);
        BWD_STAMP(10);
        BWD_STAMP(3);
        ex.each([&](int lane, int sl) {
          if (lane < n) {
            double quv[m];
#pragma unroll
            for (int l = 0; l < m; ++l) quv[l] = Q[(n + l) * QS + nm];
            double a_ = Q[lane * QS + nm];
            BWD_FENCE();
#pragma unroll
            for (int l = 0; l < m; ++l) a_ += Kc[sl][l] * qks[l];
#pragma unroll
            for (int l = 0; l < m; ++l) a_ -= 2.0 * Kc[sl][l] * quv[l];
            red[64 + lane] = a_;
          }
        });
        ex.sync();  // -K (written by the gain solve) and red before the Vxx stage / the gap stage read them
      } else
#endif
      {
      BWD_STAMP(8);
      ex.sync();
      BWD_STAMP(9);
      // Quu k, one row per lane (same order of summation as the single-lane form)
      ex.each([&](int lane, int sl) {
        if (lane < m) {
          double qrow[m], kv[m];
#pragma unroll
          for (int j = 0; j < m; ++j) {
            qrow[j] = Q[(n + (lane > j ? lane : j)) * QS + n + (lane > j ? j : lane)];
            kv[j] = kf[j];
          }
          const double kl = kf[lane];
          const double ql = Q[(n + lane) * QS + nm];  // Qu of this control (zeroed on a clamped one by the box path)
          BWD_FENCE();
          double a_ = 0;
#pragma unroll
          for (int j = 0; j < m; ++j) a_ += qrow[j] * kv[j];
          const double qk = a_ + ureg * kl;
          kf[m + lane] = qk;
          // this control's terms of the sums (a pass that fails at this knot starts over with the sums at zero)
          dgu_l[sl] += ql * kl;
          dqu_l[sl] -= kl * qk;
          qu2_l[sl] += ql * ql;
        }
      });
      ex.sync();
      BWD_STAMP(10);
      if (flag[0] != 0.0) {
        fail = true;
        BWD_KNOT_EXIT;
      }
      BWD_STAMP(3);
      // Vx = Qx + K^T Quuk - 2 K^T Qu from the lane's own column; Vxx = Qxx + (Qxu)(-K) on the matrix cores: the Qxx tiles
      // are still in the accumulators of the Q stage, A operand = Qxu (columns n.. of Q; beyond m they meet zero rows of -K)
      ex.each([&](int lane, int sl) {
        if (lane < n) {
          double qkv[m], quv[m];
#pragma unroll
          for (int l = 0; l < m; ++l) {
            qkv[l] = kf[m + l];
            quv[l] = Q[(n + l) * QS + nm];
          }
          double a_ = Q[lane * QS + nm];
          BWD_FENCE();
#pragma unroll
          for (int l = 0; l < m; ++l) a_ += Kc[sl][l] * qkv[l];
#pragma unroll
          for (int l = 0; l < m; ++l) a_ -= 2.0 * Kc[sl][l] * quv[l];
          red[64 + lane] = a_;
        }
      });
      }
      BWD_STAMP(11);
#if EMPC_BWD_MFMA4
      {
        double aopV[2][Exec::SLOTS][SM::RGN], bopV[2][Exec::SLOTS][MTN];
        auto loadV = [&](int ks, int buf) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int rg = 0; rg < SM::RGN; ++rg) aopV[buf][sl][rg] = Q[(4 * rg + li % 4) * QS + n + 4 * ks + lq];
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) bopV[buf][sl][mt] = Kn[(4 * ks + lq) * KS + 16 * mt + li];
          });
        };
        loadV(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
          if (ks + 1 < KSM) loadV(ks + 1, (ks + 1) & 1);
#pragma unroll
          for (int rg = 0; rg < SM::RGN; ++rg)
#pragma unroll
            for (int nt = 0; nt < MTN; ++nt)
              if (!SM::group_skipped(rg, nt)) ex.mfma4(aopV[ks & 1], rg, bopV[ks & 1], nt, accQ, rg / 4, nt, rg % 4);
        }
      }
#else
      {
        double aopV[2][Exec::SLOTS][MTN], bopV[2][Exec::SLOTS][MTN];
        auto loadV = [&](int ks, int buf) {
          ex.each([&](int lane, int sl) {
            const int li = lane % 16, lq = lane / 16;
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) {
              aopV[buf][sl][mt] = Q[(16 * mt + li) * QS + n + 4 * ks + lq];
              bopV[buf][sl][mt] = Kn[(4 * ks + lq) * KS + 16 * mt + li];
            }
          });
        };
        loadV(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
          if (ks + 1 < KSM) loadV(ks + 1, (ks + 1) & 1);
#pragma unroll
          for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
            for (int nt = 0; nt < MTN; ++nt)
              if (!SM::tile_skipped(mt, nt)) ex.mfma(aopV[ks & 1], mt, bopV[ks & 1], nt, accQ, mt, nt);
        }
      }
#endif
      ex.each([&](int lane, int sl) {
        const int lj = lane % 16, lq = lane / 16;
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < MTN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * mt + 4 * r < n && 16 * nt < n && !SM::tile_skipped(mt, nt)) W[(16 * mt + 4 * r + lq) * WS + 16 * nt + lj] = accQ[sl][mt][nt][r];
      });
      BWD_STAMP(12);
      ex.sync();
      BWD_STAMP(4);
      // symmetrise + regularise -> V; NaN / overflow guards of Vxx and Vx are collected per lane and reduced once
      bool badl[Exec::SLOTS];
      ex.each([&](int lane, int sl) {
        bool bad = false;
#if EMPC_BWD_R4B
        double wa[NSY], wb[NSY];
#pragma unroll
        for (int q = 0; q < NSY; ++q) {
          const int oa = sy_w[sl][q][0];
          wa[q] = W[oa < 0 ? 0 : oa];
          wb[q] = W[sy_w[sl][q][1]];
        }
        BWD_FENCE();
#pragma unroll
        for (int q = 0; q < NSY; ++q) {
          if (sy_w[sl][q][0] >= 0) {
            const bool diag = sy_v[sl][q][1] < 0;
            const double h = 0.5 * (wa[q] + wb[q]);
            const double v_ = diag ? h + xreg : h;
            V[sy_v[sl][q][0]] = v_;
            if (!diag) V[sy_v[sl][q][1]] = v_;
            bad = bad || bad_number(v_);
          }
        }
#else
        constexpr int NS = (n * n + NL - 1) / NL;
        double wa[NS], wb[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const int i = lane + q * NL, ic = i < n * n ? i : 0;
          const int rr = ic / n, cc = ic % n;
          // (EMPC_BWD_SYMTILES: an entry of a tile that was not computed is replaced by its mirror image, on both sides)
          wa[q] = W[SM::entry_skipped(rr, cc) ? cc * WS + rr : rr * WS + cc];
          wb[q] = W[SM::entry_skipped(cc, rr) ? rr * WS + cc : cc * WS + rr];
        }
        BWD_FENCE();
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const int i = lane + q * NL;
          if (i < n * n) {
            const int rr = i / n, cc = i % n;
            const double v_ = 0.5 * (wa[q] + wb[q]) + ((rr == cc) ? xreg : 0.0);
            V[rr * VS + cc] = v_;
            bad = bad || bad_number(v_);
          }
        }
#endif
        badl[sl] = bad;
      });
      ex.sync();
      BWD_STAMP(5);
      // gap contribution: Vx += Vxx f ; sums for the expected improvement
      ex.each([&](int lane, int sl) {
        if (lane >= n) return;
        double a_ = 0;
        if (infeas) {
#if EMPC_BWD_R4B
          constexpr int CH = (n + 1) / 2;  // two blocks of reads (round 3: three)
#else
          constexpr int CH = 6;
          static_assert(n % CH == 0 || true, "");
#endif
#pragma unroll
          for (int j0 = 0; j0 < n; j0 += CH) {
            double vv[CH], gg[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j)
              if (j0 + j < n) {
                vv[j] = V[lane * VS + j0 + j];
                gg[j] = rec[DM::OFF_GAP + j0 + j];
              }
            BWD_FENCE();
#pragma unroll
            for (int j = 0; j < CH; ++j)
              if (j0 + j < n) a_ += vv[j] * gg[j];
            BWD_FENCE();
          }
        }
        const double nv = red[64 + lane] + (infeas ? a_ : 0.0);
        vx[lane] = nv;
        vfo[sl] = a_;
        vxo[sl] = nv;
        if (infeas) {  // (with closed gaps every term would be an exact zero)
          const double g = rec[DM::OFF_GAP + lane];
          dgf_l[sl] -= nv * g;
          dqf_l[sl] += g * a_;
        }
        badl[sl] = badl[sl] || bad_number(nv);  // NaN, inf or >= 1e30 in Vx (crocoddyl's raiseIfNaN on max |Vx|)
      });
      BWD_STAMP(13);
      const bool badAny = ex.any([&](int lane, int sl) { return badl[sl]; });
      BWD_STAMP(14);
      if (badAny) {
        fail = true;
        BWD_KNOT_EXIT;
      }
      BWD_STAMP(6);
      if (t == 0) ex.each([&](int lane, int sl) { flush_outputs(0, lane, sl); });
#if EMPC_BWD_GLDS
      return true;
    };
    if constexpr (SM::GLDS) {
      for (int t = T - 1; t >= 0; t -= 2) {
        if (!knot(t, std::integral_constant<int, 0>{})) break;
        if (t >= 1 && !knot(t - 1, std::integral_constant<int, 1>{})) break;
      }
    } else {  // (a class without room for the second buffer: one copy of the body, records staged through registers as ever)
      for (int t = T - 1; t >= 0; --t)
        if (!knot(t, std::integral_constant<int, 0>{})) break;
    }
#else
    }
#endif
#undef BWD_KNOT_EXIT
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    if (b == 0)
      ex.each([&](int lane, int sl) {
        if (lane == 0)
          for (int i = 0; i < 15; ++i) D.dbg[16 + i] = bst[i];
      });
#endif
    ex.sync();
    if (!fail) {
      // lanes' totals -> the five sums, added in lane order (the record area is dead after the last knot)
      ex.each([&](int lane, int sl) {
        rec[lane] = dgu_l[sl];
        rec[64 + lane] = dqu_l[sl];
        rec[128 + lane] = qu2_l[sl];
        rec[192 + lane] = dgf_l[sl];
        rec[256 + lane] = dqf_l[sl];
      });
      ex.sync();
      for (int i = 0; i < m; ++i) {
        dg_u += rec[i];
        dq_u += rec[64 + i];
        qu2 += rec[128 + i];
      }
      for (int i = 0; i < n; ++i) {
        dg_f += rec[192 + i];
        dq_f += rec[256 + i];
      }
      ex.sync();
      break;
    }
    xreg *= P.prm.reg_incfactor;
    if (xreg > P.prm.reg_max) xreg = P.prm.reg_max;
    ureg = xreg;
    if (xreg == P.prm.reg_max) {
      failed_final = true;
      break;
    }
  }
  if constexpr (SM::GLDS) ex.async_wait();  // (a failed pass leaves a copy in flight: nothing may land after the wavefront is gone)
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
