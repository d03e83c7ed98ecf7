#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite output (ROCm 7.2 default format) into the small CSV summaries committed here.

    python profiles/summarize_rocpd.py stats  <results.db> <out.csv>      # --kernel-trace --stats run
    python profiles/summarize_rocpd.py pmc    <results.db> <out.csv>      # --pmc run: per-kernel mean counter values
"""
import csv
import sqlite3
import sys


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for n, c, t, a, p in rows:
            w.writerow([n, c, round(t, 3), round(a, 3), round(p, 3)])


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), max(workgroup_size), max(grid_size) "
                       "from counters_collection group by kernel_name, counter_name").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max", "mean_duration_ns_profiled", "vgpr", "agpr",
                    "sgpr", "lds_bytes", "workgroup", "grid"])
        for r in rows:
            w.writerow(r)


def pmc_window(db, kernel_substr, counter, skip, count):
    """mean of `counter` over dispatches [skip, skip + count) of the kernels whose name contains kernel_substr, in dispatch order"""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    order = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "rowid")
    rows = cur.execute(f"select value from counters_collection where kernel_name like ? and counter_name = ? order by {order}",
                       (f"%{kernel_substr}%", counter)).fetchall()
    vals = [r[0] for r in rows][skip:skip + count]
    return (sum(vals) / len(vals) if vals else None), len(vals), len(rows)


def traffic(fetch_db, write_db, out_json, batch, kv_first, kv_last, layers, warmup, steps, label):
    """profiles/attn_decode_traffic.json: HBM bytes per decode-attention launch over the TIMED steps of the bench command
    (dispatches of the warm-up steps are skipped), corrected as MI355X_MICROARCH.md prescribes for gfx950."""
    import json
    batch, kv_first, kv_last, layers, warmup, steps = map(int, (batch, kv_first, kv_last, layers, warmup, steps))
    f, nf, tf = pmc_window(fetch_db, "attn_decode_kernel", "FETCH_SIZE", warmup * layers, steps * layers)
    w, nw, tw = pmc_window(write_db, "attn_decode_kernel", "WRITE_SIZE", warmup * layers, steps * layers)
    out = {"round": 2, "kernel": "pplhip::attn_decode_kernel<8,128>", "label": label,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; profiles/collect_r02.sh) of "
                     "`python bench.py` (default steps / warm-up); mean over the dispatches of the timed steps only",
           "source_short": f"profiles/attn_decode_traffic.json ({label}; PMC FETCH_SIZE x2 + WRITE_SIZE over {nf} dispatches of the timed steps)",
           "batch": batch, "kv_len_first": kv_first, "kv_len_last": kv_last, "kv_quant": 8, "cache_mode": 0,
           "dispatches_used": nf, "dispatches_total": tf, "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w,
           "correction": "gfx950: FETCH_SIZE counts a 128-B request as 64 B for wide (16 B/lane) coalesced reads, so it is doubled "
                         "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE used as is",
           "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024) if f is not None and w is not None else None}
    json.dump(out, open(out_json, "w"), indent=1)


def pmc_series(db, kernel_substr, counter, column="value"):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    order = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "rowid")
    return [r[0] for r in cur.execute(f"select {column} from counters_collection where kernel_name like ? and counter_name = ? order by {order}",
                                      (f"%{kernel_substr}%", counter)).fetchall()]


def traffic_table(fetch_db, write_db, out_json, model, batch, kv_len, layers, total_steps, label):
    """profiles/attn_decode_traffic.json (round 3 format): HBM bytes per decode-attention launch BY KV LENGTH.  Step i of
    `bench.py --kv-len K` (warm-up included) launches the kernel `layers` times at kv length K + i + 1; the table holds the mean
    over those launches of 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 counts a 128-B read request as 64 B for 16 B/lane
    coalesced reads, MI355X_MICROARCH.md HBM section), so bench.py can quote the traffic of whichever steps it timed."""
    import json
    batch, kv_len, layers, total_steps = map(int, (batch, kv_len, layers, total_steps))
    f, w = pmc_series(fetch_db, "attn_decode_kernel", "FETCH_SIZE"), pmc_series(write_db, "attn_decode_kernel", "WRITE_SIZE")
    dur = pmc_series(fetch_db, "attn_decode_kernel", "FETCH_SIZE", "duration")   # ns per dispatch, of the profiled (FETCH_SIZE) pass
    n = min(len(f), len(w)) // layers
    table, dtable = {}, {}
    for i in range(min(n, total_steps)):
        fs, ws = f[i * layers:(i + 1) * layers], w[i * layers:(i + 1) * layers]
        table[str(kv_len + i + 1)] = int(2 * 1024 * sum(fs) / len(fs) + 1024 * sum(ws) / len(ws))
        ds = dur[i * layers:(i + 1) * layers]
        if ds:
            dtable[str(kv_len + i + 1)] = round(sum(ds) / len(ds) / 1e3, 2)
    used = dur[:min(n, total_steps) * layers]
    out = {"round": int(label.split()[-1]) if label.split()[-1].isdigit() else 0, "kernel": "pplhip::attn_decode_kernel<8,128>", "label": label, "model": model, "batch": batch, "kv_quant": 8,
           "cache_mode": 0, "dispatches": {"FETCH_SIZE": len(f), "WRITE_SIZE": len(w)}, "layers": layers,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; profiles/collect_rNN.sh) of "
                     "`python bench.py --steps 24 --warmup 3` -- a superset of the driver's --steps 20 --warmup 5 and of the defaults",
           "source_short": f"profiles/attn_decode_traffic.json ({label}; PMC FETCH_SIZE x2 + WRITE_SIZE per dispatch, mean per kv length)",
           "correction": "gfx950: FETCH_SIZE counts a 128-B request as 64 B for wide (16 B/lane) coalesced reads, so it is doubled "
                         "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE used as is",
           "mean_duration_us": round(sum(used) / len(used) / 1e3, 2) if used else None,
           "mean_duration_note": "kernel duration of the same dispatches in the FETCH_SIZE pass (every launch in the table is a uniform batch launch: "
                                 "roofline.frac can be recomputed from this file alone: algorithmic bytes / mean_duration_us)",
           "mean_duration_us_by_kv_len": dtable,
           "hbm_bytes_per_launch_by_kv_len": table}
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:])
    elif sys.argv[1] == "traffic_table":
        traffic_table(*sys.argv[2:])
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
