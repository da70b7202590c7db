// Micro-probes of the gfx950 pipelines the solver's latency-bound kernels live on (one wavefront, one workgroup):
// cycles per instruction for dependent / independent FP64 FMA chains, dependent / independent v_mfma_f64_16x16x4_f64,
// v_rsq_f64, an LDS read-after-write round trip and a dependent ds_read chain.  Build: hipcc --offload-arch=gfx950 -O3
// tools/probes/latency_probe.hip -o gpurun_out/latency_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CLK() __builtin_readcyclecounter()
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__global__ void __launch_bounds__(64) probe(double* out, unsigned long long* cyc, double seed) {
  __shared__ double lds[1024];
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = 1.0000001, z = 0.5;
  unsigned long long t0, t1;
  int k = 0;
  // 1. dependent FMA chain
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 256; ++i) x = __builtin_fma(x, y, z);
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  // 2. four independent chains
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    a0 = __builtin_fma(a0, y, z);
    a1 = __builtin_fma(a1, y, z);
    a2 = __builtin_fma(a2, y, z);
    a3 = __builtin_fma(a3, y, z);
  }
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  x = a0 + a1 + a2 + a3;
  // 3. two independent chains
  a0 = x; a1 = x + 1;
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 128; ++i) {
    a0 = __builtin_fma(a0, y, z);
    a1 = __builtin_fma(a1, y, z);
  }
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  x = a0 + a1;
  // 4. dependent MFMA chain (same accumulator)
  d4 acc = {x, x, x, x};
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  // 5. four independent MFMA accumulators
  d4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c3, 0, 0, 0);
  }
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  x = c0[0] + c1[1] + c2[2] + c3[3];
  // 6. dependent rsq chain
  double r = 2.0 + lane;
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 64; ++i) r = __builtin_amdgcn_rsq(r) + 1.5;
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  x += r;
  // 7. LDS write -> read round trips (dependent through the value)
  lds[lane] = x;
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    lds[lane + 64 * (i & 7)] = x;
    x = lds[((lane + 1) & 63) + 64 * (i & 7)] + 1.0;
  }
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  // 8. dependent LDS reads (pointer chase)
  for (int i = lane; i < 1024; i += 64) lds[i] = (double)((i * 7 + 13) & 1023);
  __syncthreads();
  int idx = lane;
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 64; ++i) idx = (int)lds[idx];
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  // 9. 16 independent LDS reads issued together, then used
  double v[16];
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = lds[(idx + i * 65) & 1023];
  FENCE();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  // 10. dependent add chain (v_add_f64)
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 256; ++i) s = s + y;
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  // 11. dependent mul chain
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 256; ++i) s = s * y;
  FENCE(); t1 = CLK(); FENCE();
  cyc[k++] = t1 - t0;
  out[lane] = x + s + idx;
}
// dependent scalar loads (s_load_dword through the constant cache): idx = table[idx], wave-uniform
typedef const __attribute__((address_space(4))) int* kint_p;
__global__ void __launch_bounds__(64) probe_sload(const int* table, int* out, unsigned long long* cyc) {
  kint_p t = (kint_p)(const void*)table;
  int idx = 0;
  FENCE(); unsigned long long t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 64; ++i) idx = t[idx];
  FENCE(); unsigned long long t1 = CLK(); FENCE();
  cyc[0] = t1 - t0;
  // second pass over the same lines (now certainly cached)
  idx = 0;
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 64; ++i) idx = t[idx];
  FENCE(); t1 = CLK(); FENCE();
  cyc[1] = t1 - t0;
  // 8 independent scalar loads, one wait
  int a[8];
  FENCE(); t0 = CLK(); FENCE();
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = t[(i * 37) & 255];
  FENCE();
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += a[i];
  FENCE(); t1 = CLK(); FENCE();
  cyc[2] = t1 - t0;
  out[threadIdx.x] = idx + sum;
}
int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, 16 * sizeof(unsigned long long));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
  unsigned long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char* name[] = {"dependent v_fma_f64 (256)", "4 independent fma chains (256 instr)", "2 independent fma chains (256 instr)",
                        "dependent v_mfma_f64_16x16x4 (64)", "4 independent mfma accumulators (64 instr)", "dependent v_rsq_f64+add (64 pairs)",
                        "LDS write->read round trip (32)", "dependent ds_read chain (64)", "16 independent ds_reads + 16 adds (1)",
                        "dependent v_add_f64 (256)", "dependent v_mul_f64 (256)"};
  const int cnt[] = {256, 256, 256, 64, 64, 64, 32, 64, 1, 256, 256};
  for (int i = 0; i < 11; ++i) printf("%-45s %8llu cycles  = %.1f per item\n", name[i], h[i], (double)h[i] / cnt[i]);
  {
    int h_t[256], *d_t, *d_o;
    for (int i = 0; i < 256; ++i) h_t[i] = (i * 67 + 5) & 255;
    hipMalloc(&d_t, sizeof(h_t));
    hipMalloc(&d_o, 64 * sizeof(int));
    hipMemcpy(d_t, h_t, sizeof(h_t), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe_sload, dim3(1), dim3(64), 0, 0, d_t, d_o, cyc);
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-45s %8llu cycles  = %.1f per item\n", "dependent s_load_dword chain (64), first pass", h[0], h[0] / 64.0);
    printf("%-45s %8llu cycles  = %.1f per item\n", "dependent s_load_dword chain (64), cached", h[1], h[1] / 64.0);
    printf("%-45s %8llu cycles\n", "8 independent s_loads + one wait", h[2]);
  }
  return 0;
}
