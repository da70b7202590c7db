// Kernel instantiation over the BAKED constants of hextilt_flying_arm_5 (csrc/baked/, tools/bake_models.py): free dynamics.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_baked_arm5() { return make_baked_table<Dims<6, 6, BakedHextiltArm5>, 0>(empc_table_6_6()); }
