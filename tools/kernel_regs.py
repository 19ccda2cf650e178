#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` logs: python tools/kernel_regs.py <log> [filter-regex]"""
import re
import sys

rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cur = None
rows = {}
for ln in open(sys.argv[1]):
    m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    short = re.sub(r"^_ZN3bcg(12_GLOBAL__N_1)?\d+", "", name)
    short = re.sub(r"EEvNS_10LatticeDev.*", "", short)
    if rx and not rx.search(short):
        continue
    print(f"{short:44s} VGPR {r.get('VGPRs'):>4} SGPR {r.get('TotalSGPRs'):>4} spillV {r.get('VGPRs Spill'):>3} spillS {r.get('SGPRs Spill'):>3} "
          f"scratch {r.get('ScratchSize [bytes/lane]'):>4} occ {r.get('Occupancy [waves/SIMD]')}")
