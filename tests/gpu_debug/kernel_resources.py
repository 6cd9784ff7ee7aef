"""Registers, spills, scratch and LDS of the kernels in the built library (code-object metadata).
usage: python tests/gpu_debug/kernel_resources.py [substring] [--lib path]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "fnft_amd", "lib", "libfnft_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

def resources(lib=LIB):
    d = tempfile.mkdtemp()
    l2 = os.path.join(d, "lib.so")
    subprocess.check_call(["cp", lib, l2])
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", l2], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
    notes = ""
    for co in sorted(f for f in os.listdir(d) if "gfx950" in f):   # one code object per translation unit
        notes += subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(d, co)], stdout=subprocess.PIPE, text=True).stdout
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            out.append(cur); cur = {}
        if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                 "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size") and k not in cur:
            cur[k] = v
    if cur.get("name"):
        out.append(cur)
    return out

if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else ""
    lib = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else LIB
    names = subprocess.run(["c++filt"], input="\n".join(r.get("name", "") for r in resources(lib)), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for r, n in zip(resources(lib), names):
        if sub in n:
            print("%-60s vgpr %4s agpr %3s sgpr %3s spill %4s scratch %5s lds %6s" % (
                n.replace("void kernel_entry<", "").split(">(")[0][:60], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"),
                r.get("vgpr_spill_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
