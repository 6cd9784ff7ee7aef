// hip_kernels_mid.hip -- kernel instantiations of group "mid" (see hip_be.h); generated list, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KMidSym<true>)
FA_INST(KMidSym<false>)
