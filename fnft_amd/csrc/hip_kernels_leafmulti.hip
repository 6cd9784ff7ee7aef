// hip_kernels_leafmulti.hip -- kernel instantiations of group "leafmulti" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KLeafMulti<1, 3>)
FA_INST(KLeafMulti<2, 3>)
FA_INST(KLeafMulti<4, 3>)
FA_INST(KLeafMulti<1, 2>)
FA_INST(KLeafMulti<2, 2>)
FA_INST(KLeafMulti<4, 2>)
