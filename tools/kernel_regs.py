#!/usr/bin/env python3
"""Print VGPR / SGPR / spill / LDS / scratch figures of every kernel in a hipcc -S (--cuda-device-only) listing."""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    name = g("name")
    name = re.sub(r"^_ZN6fresco\d+", "", name)[:60]
    print("%-62s vgpr %3s agpr %3s spill %4s sgpr %3s lds %6s scratch %5s" % (
        name, g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("sgpr_count"),
        g("group_segment_fixed_size"), g("private_segment_fixed_size")))
