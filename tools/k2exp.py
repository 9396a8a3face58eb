import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import benchmarks
r = benchmarks.k2_sweep(batches=(64, 4096), iters=20)
print(os.environ.get("VAA_LIB_PATH", "default"), [(x["B"], round(x["mean_us"], 1)) for x in r])
