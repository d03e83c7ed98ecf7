#!/usr/bin/env python3
"""Shader clock and power while a GEMM variant runs back to back for ~1.5 s each: is the K loop clock / power limited?
(round 4: the bare MFMA stream of the asm loop, the loop with its LDS-DMA refills, the product kernel)"""
import glob, os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = sys.argv[:1] + ["1024", "0"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_asm_probe.py")).read().split("for name, N, K in SHAPES:")[0]
exec(src)


def sysfs():
    out = {}
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            cur = [l for l in open(p).read().splitlines() if "*" in l]
            out["sclk"] = cur[0] if cur else "?"
        except OSError:
            pass
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            out["power_W"] = int(open(p).read()) / 1e6
        except (OSError, ValueError):
            pass
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"):
        try:
            out["freq1_MHz"] = int(open(p).read()) / 1e6
        except (OSError, ValueError):
            pass
    return out


def smi():
    try:
        o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        return o.strip()[:600]
    except Exception as e:  # noqa
        return f"rocm-smi failed: {e}"


M, N, K = 1024, 12288, 12288
x = (torch.randn(M, K, device="cuda") * 0.5).half()
w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
xz, wz = torch.zeros_like(x), torch.zeros_like(w)
sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
y = torch.empty(M, N, device="cuda", dtype=torch.float16)
libs = {"base": probe, **ablations}
calls = {n: (lambda l=l, x=x, w=w: l.pplhip_probe_linear_w8_asm(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), M, N, K, y.data_ptr(), 0)) for n, l in libs.items()}
calls["base_zero_operands"] = lambda: probe.pplhip_probe_linear_w8_asm(None, xz.data_ptr(), wz.data_ptr(), sc.data_ptr(), M, N, K, y.data_ptr(), 0)
calls["product"] = lambda: m.lib().pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y.data_ptr(), 0)
print("idle:", sysfs(), smi(), flush=True)
for n, c in calls.items():
    c(); torch.cuda.synchronize()
    t1 = timeit(c, 10)                       # us per launch
    reps = int(1.5e6 / t1)
    samples = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        c()
    e1.record()
    t0 = time.time()
    while not e1.query():
        samples.append(sysfs())
        time.sleep(0.05)
    s_mid = smi() if False else ""
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / reps * 1e3
    mid = samples[len(samples) // 4: -max(1, len(samples) // 8)] or samples
    pw = [s.get("power_W") for s in mid if s.get("power_W")]
    fq = [s.get("freq1_MHz") for s in mid if s.get("freq1_MHz")]
    print(f"{n:22s} {per:8.1f} us/launch  {2.0 * M * N * K / per / 1e6:7.1f} TFLOP/s  samples {len(samples)}  power {min(pw) if pw else '?'}..{max(pw) if pw else '?'} W"
          f"  freq1 {min(fq) if fq else '?'}..{max(fq) if fq else '?'} MHz  sclk {mid[len(mid) // 2].get('sclk') if mid else '?'}", flush=True)
