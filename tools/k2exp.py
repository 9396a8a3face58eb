import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import benchmarks
batches = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (64, 4096)
r = benchmarks.k2_sweep(batches=batches, iters=20)
print(os.path.basename(os.environ.get("VAA_LIB_PATH", "default")), [(x["B"], round(x["mean_us"], 1)) for x in r])
