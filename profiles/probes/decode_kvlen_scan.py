import sys, os, json
sys.argv=[sys.argv[0], "/dev/null"]
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("rs", "profiles/roofline_sweep.py")
rs = importlib.util.module_from_spec(spec); spec.loader.exec_module(rs)
for kv in (496, 512, 513, 520, 539, 544, 576, 608, 640, 641):
    r = rs.run(1024, kv, 32, 32, 1, iters=32, warm=4)
    print(kv, r["us_per_launch"], r["GBps"], round(r["us_per_launch"]/kv,4), flush=True)
