// hip_kernels_pair4c.hip -- kernel instantiations of group "pair4c" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KPairFft<2048, 4>)
FA_INST(KPairFft<4096, 4>)
FA_INST(KMid<4>)
