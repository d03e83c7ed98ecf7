import numpy as np, sys
sys.path.insert(0, '.')
from tests.test_gpu_model import _defer_case_logits
for (wq, kvq, batch, hkv) in [(4,8,120,2),(4,8,120,16),(8,8,120,2),(4,0,120,2),(4,8,40,2),(0,8,120,2),(4,8,120,4)]:
    res = _defer_case_logits(wq, kvq, batch, hkv=hkv)
    out = []
    for (got, want, gtok, wtok, glp, wlp, alt) in res:
        sc = max(1.0, np.abs(want).max())
        out.append((round(float(np.abs(got-want).max()/sc),5), round(float(np.abs(alt-want).max()/sc),5), int((gtok!=wtok).sum())))
    print((wq,kvq,batch,hkv), out, flush=True)
