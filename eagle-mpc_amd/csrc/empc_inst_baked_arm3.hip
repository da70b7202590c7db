// Kernel instantiation over the BAKED constants of hexacopter370_flying_arm_3 (csrc/baked/, tools/bake_models.py): free dynamics.
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_baked_arm3() { return make_baked_table<Dims<4, 6, BakedHex370Arm3>, 0>(empc_table_4_6()); }
