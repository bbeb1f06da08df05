#!/usr/bin/env python3
"""Registers / scratch / LDS per kernel from a hipcc -S listing (the amdhsa metadata at its end).  usage: kernel_regs.py file.s [filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
