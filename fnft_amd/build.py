"""Build libfnft_amd.so (hipcc, gfx950) in-tree: fnft_amd/lib/libfnft_amd.so.

Usage: python -m fnft_amd.build [--force] [-j N]
The library is the product; it has no CPU fallback.  hipcc cross-compiles without a GPU.
The HIP side is several translation units (csrc/hip_backend.hip + csrc/hip_kernels_*.hip, see
csrc/hip_be.h) compiled in parallel; only units whose object is older than the sources are rebuilt.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfnft_amd.so")
ARCH = "gfx950"

C_SOURCES = ["fnft_nsev_host.c", "fnft_kdvv_host.c", "fnft_nsev_inverse_host.c", "fnft_nsep_host.c"]


def hip_sources():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "fnft_amd.h")]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build_id():
    """Hash of the sources the library is built from (csrc/*.h, *.hip, *.c and include/fnft_amd.h): what bench.py stamps
    its lines with and profiles/tree_traffic.json its counter figures, so that a PMC figure is only ever reported next
    to a timing of the SAME build."""
    import hashlib
    h = hashlib.sha1()
    for path in sorted([os.path.join(CSRC, s) for s in hip_sources() + C_SOURCES] + headers()):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, s) for s in hip_sources() + C_SOURCES] + headers()
    return _newest(deps) > os.path.getmtime(LIB)


def build(force=False, verbose=False, defs=None, out=None, jobs=None):
    """defs / out: diagnostic variants (tests/gpu_debug/build_variant.py) -- extra -D flags and another
    output path; the product library is always build() with neither."""
    if out is None and not force and not needs_build():
        return LIB
    lib = out or LIB
    tag = os.path.splitext(os.path.basename(lib))[0]
    objdir = os.path.join(os.path.dirname(lib), "obj_" + tag) if out else LIBDIR
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    extra = list(defs) if defs else os.environ.get("FNFT_AMD_DEFS", "").split()
    hdr_time = _newest(headers())
    stamp = os.path.join(objdir, "flags.txt")
    flags_changed = (not os.path.exists(stamp)) or open(stamp).read() != " ".join(extra)
    cmds, objs = [], []
    for src in hip_sources():
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        sp = os.path.join(CSRC, src)
        if force or flags_changed or not os.path.exists(o) or os.path.getmtime(o) < max(hdr_time, os.path.getmtime(sp)):
            cmds.append([hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"] + extra + ["-c", sp, "-o", o])
    for src in C_SOURCES:
        o = os.path.join(objdir, src.replace(".c", ".o"))
        objs.append(o)
        sp = os.path.join(CSRC, src)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(hdr_time, os.path.getmtime(sp)):
            cmds.append([hipcc, "-O2", "-std=c11", "-fPIC", "-x", "c", "-c", sp, "-o", o])

    def run(c):
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c)

    n = jobs or int(os.environ.get("FNFT_AMD_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))
    with ThreadPoolExecutor(max_workers=max(1, n)) as ex:
        list(ex.map(run, cmds))
    with open(stamp, "w") as f:
        f.write(" ".join(extra))
    # -z defs: a kernel the host logic launches but no hip_kernels_*.hip instantiates fails HERE, not at load time
    link = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-z,defs", "-o", lib] + objs
    run(link)
    return lib


if __name__ == "__main__":
    j = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose=True, jobs=j))
