#!/usr/bin/env python3
"""Matrix-pipe occupancy per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass
(counter_collection.csv): MFMA-busy SIMD-cycles / (1024 SIMDs x the kernel's active cycles per XCD), with GRBM_GUI_ACTIVE
as rocprofv3 reports it (summed over the eight XCDs).  Checked against flops / duration on the large C5-shaped products
(profiles/r04_pmc_mfma_c5s_v2.csv: k_gru<4, 1> 0.74, k_gemm_rb<2, 2> 0.76).  usage: pmc_mfma.py counters.csv [out.json]"""
import collections
import csv
import json
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    rows[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in rows.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    n = len(c['GRBM_GUI_ACTIVE'])
    mfma = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / n
    gui = sum(c['GRBM_GUI_ACTIVE']) / n
    frac = mfma / (1024.0 * gui / 8.0) if gui else 0.0  # per XCD active cycles = gui / 8; 1024 SIMDs
    out[k] = dict(calls=n, mfma_busy=mfma, gui_active=gui, mfma_busy_frac=round(frac, 4))
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['gui_active'] * kv[1]['calls'])[:40]:
    print(f"{v['mfma_busy_frac']:7.3f}  calls {v['calls']:5d}  active/launch {v['gui_active'] / 8:10.0f} cyc  {k[:90]}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
