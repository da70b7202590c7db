// Kernel instantiation for (bodies, rotors, contact dynamics) = Dims<4, 6>, problems whose stages use ContactModel3D AND ContactModel6D -- one translation unit per robot class so the build parallelises.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_4_6_contact_mixed() { return make_table<Dims<4, 6>, CT_MIXED>(); }
