#!/bin/bash
# effective shader clock of the GEMM K loops on random vs zero operands: GRBM_GUI_ACTIVE (cycles the GPU was busy) / kernel duration,
# per dispatch (rocprofv3 --pmc with --kernel-trace only).  Run on the GPU box from the repo root: bash profiles/probes/gemm_power_pmc.sh
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pw
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_pw -- python $R/profiles/probes/gemm_data_power_probe.py > /tmp/prof_pw.log 2>&1
tail -9 /tmp/prof_pw.log
db=$(find /tmp/prof_pw -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
order = "dispatch_id" if "dispatch_id" in cols else "rowid"
rows = cur.execute(f"select kernel_name, value, duration from counters_collection where counter_name = 'GRBM_GUI_ACTIVE' and kernel_name like '%gemm_w8%' order by {order}").fetchall()
# the probe runs, per operand pair, 2 warm-ups + 3 x 30 timed launches of the asm kernel, then the same for the product kernel
print("dispatches", len(rows))
import itertools
grp = []
for name, grp_rows in itertools.groupby(rows, key=lambda r: r[0][:60]):
    g = list(grp_rows)
    clk = [v / d for _, v, d in g if d > 0]   # cycles per ns = GHz
    dur = [d for _, v, d in g]
    print(f"{name:60s} n={len(g):4d} duration {sum(dur)/len(dur)/1e3:8.1f} us  effective clock {sum(clk)/len(clk):.3f} GHz (min {min(clk):.3f} max {max(clk):.3f})")
PY
