// hip_kernels_multi3a.hip -- kernel instantiations of group "multi3a" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KMulti<16, 3>)
FA_INST(KMulti<32, 3>)
FA_INST(KMulti<64, 3>)
FA_INST(KMulti<128, 3>)
