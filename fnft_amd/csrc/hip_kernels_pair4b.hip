// hip_kernels_pair4b.hip -- kernel instantiations of group "pair4b" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KPairFft<512, 4>)
FA_INST(KPairFft<1024, 4>)
