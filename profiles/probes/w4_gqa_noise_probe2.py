"""Where do the W4 + grouped-query outliers of w4_gqa_noise_probe.py come from?  Per step: largest error / scale, the row that carries it,
how many rows exceed 5e-3, under a few dispatch switches (child processes: the switches are read once)."""
import numpy as np, sys, os, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from tests.test_gpu_model import _defer_case_logits
    wq, kvq, batch, hkv = [int(a) for a in sys.argv[2:6]]
    for rep in range(2):
        res = _defer_case_logits(wq, kvq, batch, hkv=hkv)
        out = []
        for (got, want, gtok, wtok, glp, wlp, alt) in res:
            sc = max(1.0, np.abs(want).max())
            e = np.abs(got - want).max(-1) / sc
            out.append((round(float(e.max()), 5), int(e.argmax()), int((e > 5e-3).sum()), round(float(np.abs(alt - want).max() / sc), 5)))
        print("   rep", rep, out, flush=True)
    sys.exit(0)
for case in [(4, 8, 120, 2), (4, 0, 120, 2)]:
    for env in [{}, {"PPLHIP_DEFER_REDUCE": "0"}, {"PPLHIP_GEMM_SPLITK": "1"}, {"PPLHIP_GEMM_GENERIC": "1"}, {"PPLHIP_ATTN_NOGQA": "1"}]:
        print(case, env, flush=True)
        r = subprocess.run([sys.executable, __file__, "child"] + [str(c) for c in case], env=dict(os.environ, **env), capture_output=True, text=True)
        print(r.stdout + r.stderr[-500:].replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory", ""), flush=True)
