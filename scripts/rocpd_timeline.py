#!/usr/bin/env python3
"""scripts/rocpd_timeline.py <results.db> [pattern] -- print the kernel dispatches of the last bench step recorded in a
rocprofv3 rocpd database (start offset, duration, scratch, LDS, grid, VGPRs, name), starting at the last dispatch whose
name matches `pattern` minus a few."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_extract_anchor_hot"
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
names = {r[0]: (r[1], r[2]) for r in cur.execute(f"select id, kernel_name, arch_vgpr_count from {ks}")}
rows = list(cur.execute(f"select kernel_id,start,end,private_segment_size,group_segment_size,grid_size_x,workgroup_size_x,queue_id from {kd} order by start"))
idx = [i for i, r in enumerate(rows) if pat in names[r[0]][0]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
start = max(0, idx[-back] - 12) if len(idx) >= back else 0
t0 = rows[start][1]
for k, s, e, p, g, gx, wx, q in rows[start:]:
    n = names[k][0]
    if "rocclr" in n and (e - s) < 8000:
        continue
    print(f"@{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:9.1f} us q{q} scr {p:4d} lds {g:6d} grid {gx // wx:6d}x{wx} vgpr {names[k][1]:3d} {n[:60]}")
