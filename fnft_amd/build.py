"""Build libfnft_amd.so (hipcc, gfx950) in-tree: fnft_amd/lib/libfnft_amd.so.

Usage: python -m fnft_amd.build [--force]
The library is the product; it has no CPU fallback.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfnft_amd.so")
ARCH = "gfx950"

SOURCES = ["hip_backend.hip", "fnft_nsev_host.c", "fnft_kdvv_host.c"]
HEADERS = ["dev_compat.h", "fft_dev.h", "nft_kernels.h", "nft_dispatch.h", "nft_plan.h", "nft_api.h", "nft_schemes.h", "nft_discspec.h",
           os.path.join("..", "..", "include", "fnft_amd.h")]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return _newest(deps) > os.path.getmtime(LIB)


def build(force=False, verbose=False, defs=None, out=None):
    """defs / out: diagnostic variants (tests/gpu_debug/build_variant.py) -- extra -D flags and another
    output path; the product library is always build() with neither."""
    if out is None and not force and not needs_build():
        return LIB
    lib = out or LIB
    libdir = os.path.dirname(lib) if out else LIBDIR
    os.makedirs(libdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    extra = list(defs) if defs else os.environ.get("FNFT_AMD_DEFS", "").split()
    tag = os.path.splitext(os.path.basename(lib))[0]
    cmds = [
        [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"] + extra + ["-c",
         os.path.join(CSRC, "hip_backend.hip"), "-o", os.path.join(libdir, tag + "_hip_backend.o" if out else "hip_backend.o")],
        [hipcc, "-O2", "-std=c11", "-fPIC", "-x", "c", "-c",
         os.path.join(CSRC, "fnft_nsev_host.c"), "-o", os.path.join(libdir, tag + "_nsev_host.o" if out else "fnft_nsev_host.o")],
        [hipcc, "-O2", "-std=c11", "-fPIC", "-x", "c", "-c",
         os.path.join(CSRC, "fnft_kdvv_host.c"), "-o", os.path.join(libdir, tag + "_kdvv_host.o" if out else "fnft_kdvv_host.o")],
    ]
    for c in cmds:
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c)
        objs.append(c[-1])
    link = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
