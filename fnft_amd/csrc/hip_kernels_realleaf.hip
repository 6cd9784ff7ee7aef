// hip_kernels_realleaf.hip -- kernel instantiations of group "realleaf" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KRLeafStrang<6, false>)
FA_INST(KRLeafStrang<6, true>)
FA_INST(KRLeafStrang<8, false>)
FA_INST(KRLeafStrang<8, true>)
