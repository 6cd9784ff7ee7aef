"""Per-kernel table from rocprofv3 --pmc counter_collection CSVs (one directory per pass).

    python profiles/pmc_table.py gpurun_out/pmc_TAG  >  profiles/rNN_pmc_table.txt

Averages every counter over the dispatches of a kernel.  Fractions are of SQ_WAVE_CYCLES (summed over
waves); fetchMB = 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md, HBM section; the
counter's unit is KiB here -- checked against streaming kernels of known size), writeMB = WRITE_SIZE.
busy_us = SQ_BUSY_CYCLES-derived; clk = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration when available.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.match(r"void kernel_entry<(.*)>\(", name)
    return m.group(1).strip() if m else name[:40]


def load(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
        per = collections.defaultdict(float)
        info = {}
        for r in csv.DictReader(open(f)):
            k = (r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])
            per[k] += float(r["Counter_Value"])
            info[r["Kernel_Name"]] = (r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
        for (did, kn, cn), v in per.items():
            acc[short(kn)][cn].append(v)
        for kn, v in info.items():
            meta[short(kn)] = v
    return acc, meta


def main():
    acc, meta = load(sys.argv[1])
    def avg(k, c):
        v = acc[k].get(c)
        return sum(v) / len(v) if v else float("nan")
    print("%-34s %5s %6s %5s | %5s %5s %5s | %5s %5s %5s %7s | %9s %9s %8s %6s | %7s %7s | %8s" % (
        "kernel", "vgpr", "lds", "scr", "wait", "winst", "activ", "valu", "lds", "wlds", "ldsconf", "i_valu", "i_lds", "i_salu", "waves",
        "fetchMB", "writeMB", "wavecyc/w"))
    for k in sorted(acc):
        wc = avg(k, "SQ_WAVE_CYCLES")
        fr = lambda c: avg(k, c) / wc if wc == wc and wc > 0 else float("nan")
        idx = avg(k, "SQ_LDS_IDX_ACTIVE")
        waves = avg(k, "SQ_WAVES")
        m = meta.get(k, ("?", "?", "?"))
        print("%-34s %5s %6s %5s | %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f %7.2f | %9.3g %9.3g %8.3g %6.0f | %7.1f %7.1f | %8.0f" % (
            k[:34], m[0], m[1], m[2], fr("SQ_WAIT_ANY"), fr("SQ_WAIT_INST_ANY"), fr("SQ_ACTIVE_INST_ANY"), fr("SQ_ACTIVE_INST_VALU"),
            fr("SQ_ACTIVE_INST_LDS"), avg(k, "SQ_WAIT_INST_LDS") / wc if wc == wc and wc > 0 else float("nan"),
            avg(k, "SQ_LDS_BANK_CONFLICT") / idx if idx == idx and idx > 0 else float("nan"),
            avg(k, "SQ_INSTS_VALU"), avg(k, "SQ_INSTS_LDS"), avg(k, "SQ_INSTS_SALU"), waves,
            2 * avg(k, "FETCH_SIZE") * 1024 / 1e6, avg(k, "WRITE_SIZE") * 1024 / 1e6,
            wc / waves if waves == waves and waves > 0 else float("nan")))
    print()
    print("busy/gui: ", end="")
    for k in sorted(acc):
        b, g = avg(k, "SQ_BUSY_CYCLES"), avg(k, "GRBM_GUI_ACTIVE")
        if b == b:
            print("%s busy=%.3g gui=%.3g; " % (k[:20], b, g), end="")
    print()


if __name__ == "__main__":
    main()
