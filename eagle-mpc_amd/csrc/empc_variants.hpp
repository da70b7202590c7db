// empc_variants.hpp -- build-time switches of the kernel bodies, all OFF in the shipped library.
//
// Rule: a change to the default build of a hot kernel is committed only together with a hardware run of
// the whole GPU suite on that tree.  A change prepared while no GPU is reachable lives behind one of the switches below, off,
// and is switched on by default only after `tools/gpu_r5.sh ab` has run it on an MI355X (suite green, timed against the
// default).  With every switch at 0 the device code of the library is the one of commit 7289ad3, the last tree whose GPU suite
// ran on hardware (gpurun_out/r04c3_pytest.log, 159 passed); tools/codeobj_compare.py checks that per kernel, byte for byte.
//
// A variant library is built next to the product:  make -C eagle-mpc_amd BUILD=build_<tag> LIB=libempc_<tag>.so EXTRA="-DEMPC_...=1"
// and loaded through EMPC_LIB_PATH.
#pragma once

// backward pass, round-4 late changes (commit 3e3ef6c; emulator + ISA verified only): symmetrisation over the upper triangle
// with per-trajectory offset tables, the record moved in 16-byte pieces, two read blocks in the gap product
#ifndef EMPC_BWD_R4B
#define EMPC_BWD_R4B 0
#endif
// BlockExec::any by wavefront ballot instead of __syncthreads_or (commit 3e3ef6c).  __syncthreads_or is also a barrier and an
// LDS fence; the ballot form adds an explicit scheduling barrier in its place (nothing may move across it).
#ifndef EMPC_ANY_BALLOT
#define EMPC_ANY_BALLOT 0
#endif
// box QP with one exit and unrolled loops (commit b94f9cd; not timed, not run on hardware)
#ifndef EMPC_BOXQP_ONE_EXIT
#define EMPC_BOXQP_ONE_EXIT 0
#endif
// fsqrt's failure value (negative or NaN argument) made from the bit pattern instead of `x == x` and `x * NaN`, which the
// -fno-honor-nans build of the baked units may fold (never observed to; prepared with the other changes that wait for hardware)
#ifndef EMPC_FSQRT_BITS
#define EMPC_FSQRT_BITS 0
#endif
