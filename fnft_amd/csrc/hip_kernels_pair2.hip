// hip_kernels_pair2.hip -- kernel instantiations of group "pair2" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KPairFft<8, 2>)
FA_INST(KPairFft<16, 2>)
FA_INST(KPairFft<32, 2>)
FA_INST(KPairFft<64, 2>)
FA_INST(KPairFft<128, 2>)
FA_INST(KPairFft<256, 2>)
FA_INST(KPairFft<512, 2>)
FA_INST(KPairFft<1024, 2>)
FA_INST(KPairFft<2048, 2>)
FA_INST(KPairFft<4096, 2>)
