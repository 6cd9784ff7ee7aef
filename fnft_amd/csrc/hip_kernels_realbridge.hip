// hip_kernels_realbridge.hip -- kernel instantiations of group "realbridge" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KRBridge<2>)
FA_INST(KRBridge<4>)
FA_INST(KRBridge<8>)
FA_INST(KRBridge<16>)
FA_INST(KRBridge<32>)
FA_INST(KRBridge<64>)
FA_INST(KRBridge<128>)
FA_INST(KRBridge<256>)
FA_INST(KRBridge<512>)
FA_INST(KRBridge<1024>)
FA_INST(KMidGen<1024>)
FA_INST(KMidGen<2048>)
