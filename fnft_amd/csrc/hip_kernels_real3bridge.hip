// hip_kernels_real3bridge.hip -- kernel instantiations of group "real3bridge" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.
#define FA_HIP_RUN_IMPL
#include "hip_be.h"

FA_INST(KR3Bridge<1>)
FA_INST(KR3Bridge<2>)
FA_INST(KR3Bridge<4>)
FA_INST(KR3Bridge<8>)
FA_INST(KR3Bridge<16>)
FA_INST(KR3Bridge<32>)
FA_INST(KR3Bridge<64>)
FA_INST(KR3Bridge<128>)
FA_INST(KR3Bridge<256>)
