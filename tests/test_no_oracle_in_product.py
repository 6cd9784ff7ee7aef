"""The oracle is test infrastructure: nothing under fnft_amd/ or include/ may mention it, and the
product library must not link it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_sources_do_not_reference_oracle_or_emulator():
    bad = []
    for base in ("fnft_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    for needle in ("import oracle", "from oracle", "fnft_oracle", "liboracle", "libfnft_emu"):
                        if needle in txt:
                            bad.append((os.path.join(dp, fn), needle))
    assert not bad, bad


def test_product_library_links_no_oracle():
    lib = os.path.join(ROOT, "fnft_amd", "lib", "libfnft_amd.so")
    if not os.path.exists(lib):
        from fnft_amd import build
        build.build()
    out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
    assert "oracle" not in out and "emu" not in out
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "orc_" not in syms
