// Kernel instantiation for (bodies, rotors, contact dynamics) = Dims<3, 6>: ContactModel3D and ContactModel6D stages behind the branch on the node's
// contact type (CT_MIXED).  No shipped file puts a contact on this robot class; the factory accepts one (src/factory/contacts.cpp:26-79).
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_3_6_contact() { return make_table<Dims<3, 6>, CT_MIXED>(); }
