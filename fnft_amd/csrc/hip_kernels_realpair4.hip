// hip_kernels_realpair4.hip -- kernel instantiations of group "realpair4" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KRPair4<32>)
FA_INST(KRPair4<64>)
FA_INST(KRPair4<128>)
FA_INST(KRPair4<256>)
FA_INST(KRPair4<512>)
