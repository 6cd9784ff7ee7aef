// hip_kernels_multi2.hip -- kernel instantiations of group "multi2" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KMulti<16, 2>)
FA_INST(KMulti<32, 2>)
FA_INST(KMulti<64, 2>)
FA_INST(KMulti<128, 2>)
FA_INST(KMulti<256, 2>)
FA_INST(KMulti<512, 2>)
FA_INST(KMulti<1024, 2>)
FA_INST(KMulti<2048, 2>)
