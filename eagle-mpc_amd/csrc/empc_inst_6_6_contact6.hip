// Kernel instantiation for (bodies, rotors, contact dynamics) = Dims<6, 6>, ContactModel6D -- one translation unit per robot class so the build parallelises.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_6_6_contact6() { return make_table<Dims<6, 6>, 6>(); }
