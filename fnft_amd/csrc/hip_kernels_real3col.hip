// hip_kernels_real3col.hip -- kernel instantiations of group "real3col" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KR3ColFwd<1>)
FA_INST(KR3ColFwd<2>)
FA_INST(KR3ColFwd<4>)
FA_INST(KR3ColFwd<8>)
FA_INST(KR3ColFwd<16>)
FA_INST(KR3ColFwd<32>)
FA_INST(KR3ColFwd<64>)
FA_INST(KR3ColFwd<128>)
FA_INST(KR3ColFwd<256>)
FA_INST(KR3ColFwd<512>)
FA_INST(KR3ColInv<1>)
FA_INST(KR3ColInv<2>)
FA_INST(KR3ColInv<4>)
FA_INST(KR3ColInv<8>)
FA_INST(KR3ColInv<16>)
FA_INST(KR3ColInv<32>)
FA_INST(KR3ColInv<64>)
FA_INST(KR3ColInv<128>)
FA_INST(KR3ColInv<256>)
FA_INST(KR3ColInv<512>)
