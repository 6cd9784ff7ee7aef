"""bench.py against a diagnostic variant of the library: python tests/gpu_debug/variant_bench.py LIB [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fnft_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
