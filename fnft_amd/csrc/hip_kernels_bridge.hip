// hip_kernels_bridge.hip -- kernel instantiations of group "bridge" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KColBridge<2>)
FA_INST(KColBridge<4>)
FA_INST(KColBridge<8>)
FA_INST(KColBridge<16>)
FA_INST(KColBridge<32>)
FA_INST(KColBridge<64>)
FA_INST(KColBridge<128>)
FA_INST(KColBridge<256>)
FA_INST(KColBridge<512>)
FA_INST(KColBridge2<2>)
FA_INST(KColBridge2<4>)
FA_INST(KColBridge2<8>)
FA_INST(KColBridge2<16>)
FA_INST(KColBridge2<32>)
FA_INST(KColBridge2<64>)
FA_INST(KColBridge2<128>)
FA_INST(KColBridge2<256>)
FA_INST(KColBridge2<512>)
