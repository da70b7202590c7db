// Kernel instantiations over the BAKED constants of the quadrotors: iris, iris_px4 (csrc/baked/, tools/bake_models.py).
#define EMPC_INSTANTIATE
#include "empc_launch.hpp"
KernelTable empc_table_baked_iris() { return make_baked_table<Dims<1, 4, BakedIris>, 0>(empc_table_1_4()); }
KernelTable empc_table_baked_iris_px4() { return make_baked_table<Dims<1, 4, BakedIrisPx4>, 0>(empc_table_1_4()); }
