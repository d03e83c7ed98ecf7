"""All rows of x equal: every output row must be identical.  Which (row, column) deviate?"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
m = load_pplhip(); L = m.lib()
torch.manual_seed(0)
N, K = 12288, 4096
M = int(sys.argv[1]) if len(sys.argv) > 1 else 254
w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda"); sc = (torch.rand(N, device="cuda") * 0.001).half()
for trial in range(3):
    row = (torch.randn(1, K, device="cuda") * 0.5).half()
    x = row.repeat(M, 1).contiguous()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    assert L.pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 0, M, N, K, y.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    ref = y[0:1]
    d = (y != ref)
    rows = torch.nonzero(d.any(1)).flatten().tolist()
    cols = torch.nonzero(d.any(0)).flatten().tolist()
    print(f"trial {trial}: rows deviating from row 0: {len(rows)} {rows[:48]}; columns {len(cols)} {cols[:16]}")
    if cols:
        c = cols[0]
        vals, counts = torch.unique(y[:, c].float(), return_counts=True)
        print("   column", c, "values", vals.tolist(), "counts", counts.tolist(), " exact:", float((row[0].double() * (w[c].double())).sum() * sc[c].double()))
