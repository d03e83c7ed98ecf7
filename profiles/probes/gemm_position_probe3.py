"""All rows of x equal, fp32 output: compare the raw fp32 results of the rows."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
m = load_pplhip(); L = m.lib()
torch.manual_seed(0)
N, K, M = 12288, 4096, 256
w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda"); sc = (torch.rand(N, device="cuda") * 0.001).half()
row = (torch.randn(1, K, device="cuda") * 0.5).half()
x = row.repeat(M, 1).contiguous()
y = torch.empty(M, N, device="cuda", dtype=torch.float32)
assert L.pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 0, M, N, K, y.data_ptr(), 1) == 0
torch.cuda.synchronize()
d = (y != y[0:1])
print("rows deviating:", torch.nonzero(d.any(1)).flatten().tolist()[:40])
print("fraction of columns deviating per deviating row:", d[112].float().mean().item(), d[127].float().mean().item())
rel = ((y[112] - y[0]).abs() / y[0].abs().clamp_min(1e-9))
print("relative deviation: median of nonzero", rel[rel > 0].median().item(), "max", rel.max().item())
i8 = y.view(torch.int32)
print("ulp distance histogram row 112 vs row 0:", torch.unique((i8[112] - i8[0]).abs().clamp_max(8), return_counts=True))
