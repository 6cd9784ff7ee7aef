// hip_kernels_multi3b.hip -- kernel instantiations of group "multi3b" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KMulti<256, 3>)
FA_INST(KMulti<512, 3>)
FA_INST(KMulti<1024, 3>)
