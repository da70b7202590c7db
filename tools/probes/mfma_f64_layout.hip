// Probe of the operand layout of v_mfma_f64_16x16x4_f64 on gfx950 (used by the MFMA form of the backward pass):
//   hipcc --offload-arch=gfx950 tools/probes/mfma_f64_layout.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Expected (and asserted): A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16],
//                          D[i = 4 * r + lane / 16][j = lane % 16] for the 4 result registers r.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];   // A is 16 x 4 row-major
  const double b = B[(l / 16) * 16 + (l % 16)];  // B is 4 x 16 row-major
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * r + l / 16) * 16 + (l % 16)] = c[r];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) {
    hA[i] = std::sin(0.37 * i) + 0.1 * i;
    hB[i] = std::cos(0.11 * i) - 0.05 * i;
  }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
      ref[i * 16 + j] = s;
    }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 256; ++i) err = std::fmax(err, std::fabs(hD[i] - ref[i]));
  std::printf("max |D - A B| = %.3e  (%s)\n", err, err < 1e-12 ? "layout confirmed" : "LAYOUT MISMATCH");
  return err < 1e-12 ? 0 : 1;
}
