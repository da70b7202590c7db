// Kernel instantiation for (bodies, rotors, ContactModel6D dynamics) = Dims<4, 6> -- one translation unit per robot class so the build parallelises.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_4_6_contact6() { return make_table<Dims<4, 6>, 6>(); }
