// hip_kernels_mid.hip -- kernel instantiations of group "mid" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KMidSym<true>)
FA_INST(KMidSym<false>)
