#!/usr/bin/env python3
"""Summarise gpurun_out/w8_ranges_timeline.log (profiles/probes/w8_ranges_timeline.sh): per wave, the cycles between the stamps of the
first iterations.  Multiplying waves (0-3): top -> barrier passed -> MFMAs issued; issuing waves (4-7): own pieces landed -> barrier -> issued."""
import re, sys
for line in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/w8_ranges_timeline.log"):
    if not line.startswith("[rg_time]"):
        print(line.strip()[:200]); continue
    m = re.match(r"\[rg_time\] M=(\d+) N=(\d+) K=(\d+) B=(\d+) units/block=(\d+) block (\d+) wave (\d+): (.*)", line)
    M, N, K, B, nu, b, wv = map(int, m.groups()[:7]); v = [0] + list(map(int, m.group(8).split()))
    if wv not in (0, 4, 6): continue
    it = []
    for t in range(min(nu, 10)):
        base = 2 + 3 * t
        if base + 2 < len(v):
            prev = v[base - 1]
            it.append((v[base] - prev, v[base + 1] - v[base], v[base + 2] - v[base + 1]))
    e = 2 + 3 * nu
    print(f"M={M} N={N} K={K} nu={nu} blk{b} w{wv}: first {v[1]} | " + " ".join(f"{a}/{bb}/{c}" for a, bb, c in it) +
          f" | loop_end {v[e] if e < len(v) else -1} end {v[e + 1] if e + 1 < len(v) else -1}")
