#!/usr/bin/env python3
"""Register / scratch / LDS summary of the kernels in a device assembly file (the .amdhsa metadata at its end):
   python tools/kernel_regs.py kzg_amd/build/ntt_dev_pp.s [name-filter]"""
import re, sys
text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in text.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print("%-110s vgpr %s agpr %s sgpr %s spill %s scratch %s" % (name[:110], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
