#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of the library's kernels, from the assembly hipcc leaves behind with -save-temps.
usage: python tools/kernel_regs.py [filter ...]   (compiles mrh_capi.hip into /tmp/mrh_regs; ~40 s)"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrhash_amd import build as b  # noqa: E402

out = os.environ.get("MRH_REGS_DIR", "/tmp/mrh_regs")
os.makedirs(out, exist_ok=True)
flags = [f for f in b.HIPCC_FLAGS if not f.startswith("-W")] + os.environ.get("MRH_EXTRA_FLAGS", "").split()  # e.g. MRH_EXTRA_FLAGS="-DMRH_BACK_WAVES=5"
subprocess.run([b.hipcc()] + flags + ["-save-temps", "-o", os.path.join(out, "x.so"), os.path.join(b.CSRC, "mrh_capi.hip")], cwd=out, check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
txt = open(glob.glob(os.path.join(out, "*gfx950*.s"))[0]).read()
md = txt[txt.index("amdhsa.kernels:"):]
want = sys.argv[1:] or ["k_back", "k_front", "k_mc", "k_scan"]
filt = "c++filt" if subprocess.run(["which", "c++filt"], capture_output=True).returncode == 0 else None
for blk in md.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dn = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
    dn = dn.split("(")[0].replace("void ", "")
    if not any(w in dn for w in want):
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)  # noqa: E731
    print(f"{dn[:88]:88s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s}")
