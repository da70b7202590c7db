// Kernel instantiation for (bodies, rotors, contact dynamics) = Dims<6, 6>, problems with a stage whose ContactModelMultiple holds TWO
// ContactModel3D contacts (CT_PAIR3: six stacked rows, one KKT system; src/stage.cpp:38-48 adds every name of the stage's list).
// No shipped file lists more than one contact; opt-in (EMPC_EXPERIMENTAL_CONTACT) until it has run on hardware.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_6_6_contact_pair() { return make_table<Dims<6, 6>, CT_PAIR3>(); }
