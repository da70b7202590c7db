// Kernel instantiations over the BAKED constants of the six-rotor platforms without an arm: hexacopter370, hextilt
// (csrc/baked/, tools/bake_models.py).
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_baked_hex370() { return make_baked_table<Dims<1, 6, BakedHex370>, 0>(empc_table_1_6()); }
KernelTable empc_table_baked_hextilt() { return make_baked_table<Dims<1, 6, BakedHextilt>, 0>(empc_table_1_6()); }
