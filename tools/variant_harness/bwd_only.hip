// Compile harness of tools/variant_verdicts.py: the backward-pass kernels alone (3 s instead of the 3 minutes of the library), so
// that a build-time variant (csrc/empc_variants.hpp) can be judged on its registers, spills, scratch and the instruction mix of the
// knot loop:   hipcc -O3 -std=c++17 -I include -I eagle-mpc_amd/csrc --offload-arch=gfx950 -D<switch>=1 -c tools/variant_harness/bwd_only.hip
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
using namespace empc;
template __global__ void k_backward4<Dims<4, 6>, false>(DevBuffers);  // 9-DoF arm (north star), squash-box solver
template __global__ void k_backward4<Dims<4, 6>, true>(DevBuffers);   // ... box solvers (SolverBoxFDDP / SolverBoxDDP)
template __global__ void k_backward4<Dims<6, 6>, false>(DevBuffers);  // 11-DoF arm (configs[3])
template __global__ void k_backward4<Dims<6, 6>, true>(DevBuffers);
