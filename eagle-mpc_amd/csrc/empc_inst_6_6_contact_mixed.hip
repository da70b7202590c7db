// Kernel instantiation for (bodies, rotors, contact dynamics) = Dims<6, 6>, problems whose stages use ContactModel3D AND ContactModel6D (CT_MIXED: the
// bodies of the two single-type instantiations behind a branch on the node's contact type).  No shipped file mixes them on this robot; the factory accepts
// it (src/factory/contacts.cpp:26-79).  Opt-in at run time (EMPC_EXPERIMENTAL_CONTACT=1, empc_solver.hip find_table) until it has run on hardware.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_6_6_contact_mixed() { return make_table<Dims<6, 6>, CT_MIXED>(); }
