#!/usr/bin/env python3
"""Writes hip_kernels_*.hip: the explicit instantiations of HipBackend::run<K> (hip_be.h), grouped so that the
translation units compile in parallel in comparable time.  The lists mirror the dispatch switches of
nft_dispatch.h (FA_FOR_EACH_N1, FA_FOR_EACH_BRIDGE_N1, FA_FOR_EACH_CHIRP_N1, dispatch_multi, ...); a kernel that
is launched but missing here fails at link time (-z defs).  Run after changing a dispatch list:
    python fnft_amd/csrc/gen_kernel_units.py
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
N1_ALL = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
N1_CHIRP = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
N1_BR = [2, 4, 8, 16, 32, 64, 128, 256, 512]
PF = [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]

groups = {}
g = ["KCoeffs<%d>" % d for d in (1, 2, 3, 4)]
g += ["KCoeffsProg", "KResamplePhase", "KResampleCombine", "KBandCheck", "KBsChunk<false>", "KBsChunk<true>", "KBsCombine<false>",
      "KBsCombine<true>", "KBsPhi", "KBsMetric", "KBsPick", "KBsPsi", "KBsMatrix", "KGridMark", "KGridMarkPh", "KCompactCount", "KCompactScatter", "KInvOp", "KInvSolitons", "KInvCdt", "KPeelImport", "KPeelLeaf", "KPeelProduct<512>", "KPeelProduct<1024>", "KPeelProduct<2048>", "KAberthNewton", "KAberthSum", "KAberthApply", "KFinalizeScales", "KExportTm", "KImportLevel0"]
g += ["KLeaf<%d>" % d for d in (1, 2, 3, 4)] + ["KPairSchool<%d>" % d for d in (1, 2, 3)]
groups["misc"] = g
groups["pair4a"] = ["KPairFft<%d, 4>" % n for n in PF if n <= 256]
groups["pair4b"] = ["KPairFft<%d, 4>" % n for n in PF if 256 < n <= 1024]
groups["pair4c"] = ["KPairFft<%d, 4>" % n for n in PF if n > 1024] + ["KMid<4>"]
groups["pair2"] = ["KPairFft<%d, 2>" % n for n in PF]
groups["multi3a"] = ["KMulti<%d, 3>" % n for n in (16, 32, 64, 128)]
groups["multi3b"] = ["KMulti<%d, 3>" % n for n in (256, 512, 1024)]
groups["multi2"] = ["KMulti<%d, 2>" % n for n in (16, 32, 64, 128, 256, 512, 1024, 2048)]
groups["leafmulti"] = ["KLeafMulti<%d, %d>" % (d, s) for s in (3, 2) for d in (1, 2, 4)]
groups["mid"] = ["KMidSym<true>", "KMidSym<false>"]
groups["col"] = ["KColFwd<%d>" % n for n in N1_ALL] + ["KColInv<%d>" % n for n in N1_ALL]
groups["bridge"] = ["KColBridge<%d>" % n for n in N1_BR] + ["KColBridge2<%d>" % n for n in N1_BR + [1024, 2048, 4096]]
groups["realpair"] = ["KRealCheck", "KRCoeffsStrang<6, false>", "KRCoeffsStrang<6, true>", "KRCoeffsStrang<8, false>", "KRCoeffsStrang<8, true>"] + ["KRPairSchool<%d>" % d for d in (1, 2, 3)] + ["KRPair<%d>" % m for m in (4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048)]
groups["realpair4"] = ["KRPair4<%d>" % m for m in (32, 64, 128, 256, 512)]
N1_RCOL = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]
groups["realcol"] = ["KRColFwd<%d>" % n for n in N1_RCOL] + ["KRColInv<%d>" % n for n in N1_RCOL]
groups["realbridge"] = ["KRBridge<%d>" % n for n in N1_BR + [1024]] + ["KMidGen<1024>", "KMidGen<2048>"]
K3 = [1, 2, 4, 8, 16, 32, 64, 128, 256]
groups["real3col"] = ["KR3ColFwd<%d>" % k for k in K3 + [512]] + ["KR3ColInv<%d>" % k for k in K3 + [512]]
groups["real3bridge"] = ["KR3Bridge<%d>" % k for k in K3]
groups["realleaf"] = ["KRLeafStrang<6, false>", "KRLeafStrang<6, true>", "KRLeafStrang<8, false>", "KRLeafStrang<8, true>"]
groups["chirpa"] = ["KChirpRows"] + ["KChirpColFwd<%d, false>" % n for n in N1_CHIRP] + ["KChirpColFwd<%d, true>" % n for n in N1_CHIRP]
groups["chirpb"] = (["KChirpColInv<%d, false, false>" % n for n in N1_CHIRP] + ["KChirpColInv<%d, true, false>" % n for n in N1_CHIRP]
                    + ["KChirpColInv<%d, false, true>" % n for n in N1_CHIRP])

for name, ks in groups.items():
    with open(os.path.join(HERE, "hip_kernels_%s.hip" % name), "w") as f:
        f.write("// hip_kernels_%s.hip -- kernel instantiations of group \"%s\" (see hip_be.h); written by gen_kernel_units.py, gfx950 only.\n"
                % (name, name))
        f.write("#define FA_HIP_RUN_IMPL\n#include \"hip_be.h\"\n\n")
        for k in ks:
            f.write("FA_INST(%s)\n" % k)
print(sum(len(v) for v in groups.values()), "kernels in", len(groups), "units")
