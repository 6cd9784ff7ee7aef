// hip_kernels_realpair.hip -- kernel instantiations of group "realpair" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KRealCheck)
FA_INST(KRCoeffsStrang<6, false>)
FA_INST(KRCoeffsStrang<6, true>)
FA_INST(KRCoeffsStrang<8, false>)
FA_INST(KRCoeffsStrang<8, true>)
FA_INST(KRPairSchool<1>)
FA_INST(KRPairSchool<2>)
FA_INST(KRPairSchool<3>)
FA_INST(KRPair<4>)
FA_INST(KRPair<8>)
FA_INST(KRPair<16>)
FA_INST(KRPair<32>)
FA_INST(KRPair<64>)
FA_INST(KRPair<128>)
FA_INST(KRPair<256>)
FA_INST(KRPair<512>)
FA_INST(KRPair<1024>)
FA_INST(KRPair<2048>)
