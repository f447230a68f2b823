#!/usr/bin/env python3
"""Register / LDS / scratch statistics of the kernels in a csrc/*.hip file (device-only -S compile for gfx950).
usage: python tools/isa_stats.py conv.hip [name-filter]   (writes /tmp/<stem>.s)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "dpmn_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/%s.s" % os.path.splitext(os.path.basename(src))[0]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
                "--offload-device-only", "-S", "-o", out, src] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
meta = s[s.index("amdhsa.kernels"):]
for e in meta.split("  - .agpr_count")[1:]:
    n = re.search(r"\.name:\s+(\S+)", e).group(1)
    dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    dn = dn[dn.find("k_"):] if "k_" in dn else dn
    if flt and flt not in dn:
        continue
    g = lambda k: (re.search(k + r":\s+(\d+)", e) or [None, "?"])[1]
    print("%-110s vgpr %3s spill %3s sgpr %3s lds %6s scratch %4s" % (dn[:110], g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_count"),
                                                                    g(r"\.group_segment_fixed_size"), g(r"\.private_segment_fixed_size")))
