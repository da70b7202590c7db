// Kernel instantiation for (bodies, rotors) = Dims<1, 6> -- one translation unit per robot class so the build parallelises.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_1_6() { return make_table<Dims<1, 6>, 0>(); }
