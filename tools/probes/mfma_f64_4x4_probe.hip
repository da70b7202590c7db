// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950 (candidate for the backward pass: n = 18 -> 20 and n + m = 27 -> 28 pad to 90 % full
// 4 x 4 tiles, the 16 x 16 x 4 tiles in use are 43 % full).  Nothing in the product uses the instruction
// yet: its operand layout is taken from the ISA manual only, and this program DISCOVERS it on the chip instead of asserting it.
//   hipcc --offload-arch=gfx950 tools/probes/mfma_f64_4x4_probe.hip -o /tmp/mfma4_probe && /tmp/mfma4_probe
// Part 1 (layout): for every pair (la, lb) the A operand is 1 in lane la only and the B operand is 1 in lane lb only; the output
//   lanes that come back non-zero say which (A lane, B lane) pairs meet in which D lane.  From the 64 x 64 table the program
//   derives block(l), row(l), k(l) of the A operand, k(l), col(l) of the B operand and (block, row, col) of the D lane, checks that
//   the table is exactly that of four independent 4 x 4 x 4 products, and prints the maps.
// Part 2 (numbers): random operands through the derived maps against a double loop on the host.
// Part 3 (cost): cycles per instruction of a dependent chain and of four independent chains, next to v_mfma_f64_16x16x4_f64 and
//   v_fma_f64 (s_memtime around 256 instructions, one wavefront).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_pairs(unsigned long long* hit) {  // hit[la * 64 + lb] = mask of D lanes that are non-zero
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(d != 0.0);
      if (l == 0) hit[la * 64 + lb] = m;
    }
}
__global__ void k_apply(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
template <int MODE>
__global__ void k_cost(unsigned long long* cycles, double* sink) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  double4_t q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0}, q2 = {0, 0, 0, 0}, q3 = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    if (MODE == 0) {  // dependent 4x4x4
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
    } else if (MODE == 1) {  // four independent 4x4x4
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
    } else if (MODE == 2) {  // four independent 16x16x4
      q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, q1, 0, 0, 0);
      q2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, q2, 0, 0, 0);
      q3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, q3, 0, 0, 0);
    } else {  // four independent vector FMAs
      d0 = __builtin_fma(a, b, d0);
      d1 = __builtin_fma(a, b, d1);
      d2 = __builtin_fma(a, b, d2);
      d3 = __builtin_fma(a, b, d3);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (l == 0) cycles[MODE] = t1 - t0;
  sink[MODE * 64 + l] = d0 + d1 + d2 + d3 + q0[0] + q1[1] + q2[2] + q3[3];
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 2;                                                                \
    }                                                                          \
  } while (0)

int main() {
  unsigned long long* dh;
  CK(hipMalloc(&dh, 64 * 64 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_pairs, dim3(1), dim3(64), 0, 0, dh);
  std::vector<unsigned long long> hit(64 * 64);
  CK(hipMemcpy(hit.data(), dh, hit.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  // every (la, lb) pair meets in at most ... lanes; for a 4 x 4 x 4 x 4-block product A[b][i][k] meets B[b][k][j] in D[b][i][j]: one lane
  int multi = 0, partners_min = 64, partners_max = 0;
  for (int la = 0; la < 64; ++la) {
    int partners = 0;
    for (int lb = 0; lb < 64; ++lb) {
      const unsigned long long m = hit[la * 64 + lb];
      if (m) ++partners;
      if (m & (m - 1)) ++multi;
    }
    partners_min = partners < partners_min ? partners : partners_min;
    partners_max = partners > partners_max ? partners : partners_max;
  }
  std::printf("partners of an A lane among the B lanes: %d .. %d (expected 4: the four columns j of its block and k); pairs that land in "
              "more than one D lane: %d (expected 0)\n", partners_min, partners_max, multi);
  // derive the maps.  Two A lanes share (block, k) iff they have the same set of B partners; they share (block, row) iff ... the D lanes
  // they reach with a common partner differ only by the row.  Canonical numbering: walk the lanes in order.
  auto dlane = [&](int la, int lb) {
    const unsigned long long m = hit[la * 64 + lb];
    return m ? __builtin_ctzll(m) : -1;
  };
  int a_grp[64], b_grp[64];  // group = (block, k): A lanes with the same partner set; B lanes with the same partner set
  int ng = 0;
  for (int la = 0; la < 64; ++la) {
    a_grp[la] = -1;
    for (int lp = 0; lp < la; ++lp) {
      bool same = true;
      for (int lb = 0; lb < 64 && same; ++lb) same = (hit[la * 64 + lb] != 0) == (hit[lp * 64 + lb] != 0);
      if (same) {
        a_grp[la] = a_grp[lp];
        break;
      }
    }
    if (a_grp[la] < 0) a_grp[la] = ng++;
  }
  for (int lb = 0; lb < 64; ++lb) {
    b_grp[lb] = -1;
    for (int la = 0; la < 64; ++la)
      if (hit[la * 64 + lb]) {
        b_grp[lb] = a_grp[la];
        break;
      }
  }
  std::printf("(block, k) groups: %d (expected 16)\n", ng);
  std::printf("lane :  A group  B group  | D lane reached with the first / second / third / fourth B partner\n");
  for (int la = 0; la < 64; ++la) {
    std::printf("%4d : %7d %8d  |", la, a_grp[la], b_grp[la]);
    for (int lb = 0; lb < 64; ++lb)
      if (hit[la * 64 + lb]) std::printf("  B%-2d->D%-2d", lb, dlane(la, lb));
    std::printf("\n");
  }
  // Hypotheses.  [2] is the one EMPC_BWD_MFMA4 (csrc/empc_backward4.hpp) and the lane emulator (tests/csrc/lane_emulator.cpp mfma4) are
  // written to: the block index rides on the low 16 lanes like the row / column index of the 16 x 16 x 4 form, k (A, B) and the row
  // (D) on lane / 16.  [0] / [1]: the block on lane / 16.  If [2] is not the confirmed one, only the index maps of those two places
  // change (and the variant loses the property that W's accumulators are the B operand of the next product unless D's row and
  // B's k still share lane / 16).
  const char* names[3] = {"A: l = 16 b + 4 k + i, B: l = 16 b + 4 k + j, D: l = 16 b + 4 i + j",
                          "A: l = 16 b + 4 k + i, B: l = 16 b + 4 k + j, D: l = 16 b + 4 j + i",
                          "A: l = 16 k + 4 b + i, B: l = 16 k + 4 b + j, D: l = 16 i + 4 b + j  (the model of EMPC_BWD_MFMA4)"};
  auto amap = [](int h, int b_, int i, int k) { return h == 2 ? 16 * k + 4 * b_ + i : 16 * b_ + 4 * k + i; };
  auto dmap = [](int h, int b_, int i, int j) { return h == 2 ? 16 * i + 4 * b_ + j : (h == 0 ? 16 * b_ + 4 * i + j : 16 * b_ + 4 * j + i); };
  int confirmed = -1;
  for (int h = 0; h < 3; ++h) {
    bool ok = true;
    int want[64][64];
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb) want[la][lb] = -1;
    for (int b_ = 0; b_ < 4; ++b_)
      for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) want[amap(h, b_, i, k)][amap(h, b_, j, k)] = dmap(h, b_, i, j);
    for (int la = 0; la < 64 && ok; ++la)
      for (int lb = 0; lb < 64 && ok; ++lb) ok = dlane(la, lb) == want[la][lb] && !(hit[la * 64 + lb] & (hit[la * 64 + lb] - 1));
    std::printf("hypothesis %d (%s): %s\n", h, names[h], ok ? "CONFIRMED" : "no");
    if (ok) confirmed = h;
  }
  std::printf("layout model of EMPC_BWD_MFMA4: %s\n", confirmed == 2 ? "CONFIRMED" : "NOT CONFIRMED -- do not adopt the variant before its index maps are rewritten");
  // numbers through the confirmed map
  int rc = confirmed >= 0 ? 0 : 1;
  if (confirmed >= 0) {
    double ha[64], hb[64], hc[64], hd[64], ref[64];
    for (int l = 0; l < 64; ++l) {
      ha[l] = std::sin(0.37 * l) + 0.1 * l;
      hb[l] = std::cos(0.11 * l) - 0.05 * l;
      hc[l] = 0.01 * l;
    }
    for (int b = 0; b < 4; ++b)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const int ld = dmap(confirmed, b, i, j);
          double s = hc[ld];
          for (int k = 0; k < 4; ++k) s += ha[amap(confirmed, b, i, k)] * hb[amap(confirmed, b, j, k)];
          ref[ld] = s;
        }
    double *da, *db, *dc, *dd;
    CK(hipMalloc(&da, sizeof(ha))); CK(hipMalloc(&db, sizeof(hb))); CK(hipMalloc(&dc, sizeof(hc))); CK(hipMalloc(&dd, sizeof(hd)));
    CK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_apply, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    CK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
    double err = 0;
    for (int l = 0; l < 64; ++l) err = std::fmax(err, std::fabs(hd[l] - ref[l]));
    std::printf("random operands through the confirmed map: max |D - (C + A B)| = %.3e (%s; the order of the four k terms inside the "
                "instruction decides the last bit)\n", err, err < 1e-12 ? "ok" : "MISMATCH");
    if (!(err < 1e-12)) rc = 1;
  }
  // cost
  unsigned long long* dcy;
  double* dsink;
  CK(hipMalloc(&dcy, 4 * sizeof(unsigned long long)));
  CK(hipMalloc(&dsink, 4 * 64 * sizeof(double)));
  for (int rep = 0; rep < 2; ++rep) {  // second pass: warm instruction cache
    hipLaunchKernelGGL(k_cost<0>, dim3(1), dim3(64), 0, 0, dcy, dsink);
    hipLaunchKernelGGL(k_cost<1>, dim3(1), dim3(64), 0, 0, dcy, dsink);
    hipLaunchKernelGGL(k_cost<2>, dim3(1), dim3(64), 0, 0, dcy, dsink);
    hipLaunchKernelGGL(k_cost<3>, dim3(1), dim3(64), 0, 0, dcy, dsink);
  }
  unsigned long long cy[4];
  CK(hipMemcpy(cy, dcy, sizeof(cy), hipMemcpyDeviceToHost));
  const char* what[4] = {"v_mfma_f64_4x4x4, dependent chain", "v_mfma_f64_4x4x4, four independent", "v_mfma_f64_16x16x4, four independent",
                         "v_fma_f64, four independent"};
  for (int m = 0; m < 4; ++m)
    std::printf("%-40s %8.1f s_memtime ticks per instruction (256 instructions, one wavefront)\n", what[m],
                (double)cy[m] / 256.0);
  return rc;
}
