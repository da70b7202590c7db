// Kernel instantiation over the BAKED constants of hexacopter680_flying_arm_2 (csrc/baked/, tools/bake_models.py): free dynamics.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_baked_arm2() { return make_baked_table<Dims<3, 6, BakedHex680Arm2>, 0>(empc_table_3_6()); }
