import json, os, sys
sys.path.insert(0, os.getcwd())
from roboticattack_amd import benchmarks
r = benchmarks.rank_shapes()
for cfg, d in r.items():
    for k, v in d.items():
        if isinstance(v, dict) and "mean_us" in v:
            print(cfg, k, round(v["mean_us"], 1))
