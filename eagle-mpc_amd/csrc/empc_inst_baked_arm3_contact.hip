// Kernel instantiation over the BAKED constants of hexacopter370_flying_arm_3 (csrc/baked/, tools/bake_models.py): ContactModel3D rows.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_baked_arm3_contact() { return make_baked_table<Dims<4, 6, BakedHex370Arm3>, 3>(empc_table_4_6_contact()); }
