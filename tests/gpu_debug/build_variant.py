"""Diagnostic build of the library with extra -D flags into build/variants/libfnft_amd_NAME.so
(git-ignored; travels to the GPU box).  usage: python tests/gpu_debug/build_variant.py NAME -DFOO=1 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fnft_amd import build as B
name, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "build", "variants", "libfnft_amd_%s.so" % name)
print(B.build(force=True, verbose=False, defs=defs, out=out))
