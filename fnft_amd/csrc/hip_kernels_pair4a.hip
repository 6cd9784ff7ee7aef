// hip_kernels_pair4a.hip -- kernel instantiations of group "pair4a" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KPairFft<8, 4>)
FA_INST(KPairFft<16, 4>)
FA_INST(KPairFft<32, 4>)
FA_INST(KPairFft<64, 4>)
FA_INST(KPairFft<128, 4>)
FA_INST(KPairFft<256, 4>)
