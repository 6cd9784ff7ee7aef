"""ISA summary of one kernel of the built library: the order of global loads / stores, s_waitcnt vmcnt and
s_barrier, run-length compressed -- shows whether loads are issued back to back or one round trip at a time.
usage: python tests/gpu_debug/isa_summary.py 'KMid<2>' [--full]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "fnft_amd", "lib", "libfnft_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

def disasm(lib=LIB):
    d = tempfile.mkdtemp()
    l2 = os.path.join(d, "lib.so")
    subprocess.check_call(["cp", lib, l2])
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", l2], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
    out = ""
    for co in sorted(f for f in os.listdir(d) if "gfx950" in f):   # one code object per translation unit
        out += subprocess.run([LLVM + "/llvm-objdump", "-d", os.path.join(d, co)], stdout=subprocess.PIPE, text=True).stdout
    dm = subprocess.run(["c++filt"], input=out, stdout=subprocess.PIPE, text=True).stdout
    return dm

def kernels(text):
    cur, res = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            res[cur] = []
        elif cur is not None:
            res[cur].append(line)
    return res

if __name__ == "__main__":
    want = sys.argv[1]
    ks = kernels(disasm(sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--lib" else LIB))
    for name, body in ks.items():
        if ("kernel_entry<" + want + " >") not in name and ("kernel_entry<" + want + ">") not in name:
            continue
        ev = []
        for l in body:
            t = l.split("//")[0].strip()
            op = t.split()[0] if t else ""
            if op.startswith(("global_load", "buffer_load")): ev.append("L")
            elif op.startswith(("global_store", "buffer_store")): ev.append("S")
            elif op.startswith("global_atomic"): ev.append("A")
            elif op == "s_barrier": ev.append("|")
            elif op == "s_waitcnt" and "vmcnt" in t: ev.append("w%s" % re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif op.startswith("scratch_"): ev.append("x")
            elif op.startswith("ds_"): ev.append("d")
            elif op.startswith("v_"): ev.append("v")
        # run-length compress
        out, i = [], 0
        while i < len(ev):
            j = i
            while j < len(ev) and ev[j] == ev[i]: j += 1
            out.append(ev[i] + (str(j - i) if j - i > 1 else ""))
            i = j
        print(name, "instructions:", len(body))
        print(" ".join(out))
