#!/usr/bin/env bash
# in-step A/B of the fused LM head: rocprofv3 kernel stats of the bench step with VAA_FUSED_HEAD=0 / 1 at bs=64 and bs=8 (traces stay in /tmp)
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"
out=$root/gpurun_out/r4h; mkdir -p $out
export TMPDIR=/tmp
for bs in 64; do
for v in 0 1; do
  VAA_FUSED_HEAD=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h${bs}_$v -o h -- python bench.py --bs $bs --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank --profile-steps 0 > $out/bench_bs${bs}_$v.json 2> $out/bench_bs${bs}_$v.err
  f=$(find /tmp/prof_h${bs}_$v -name '*kernel_stats.csv' | head -1)
  echo "== bs=$bs VAA_FUSED_HEAD=$v  $(python -c "import json;d=json.load(open('$out/bench_bs${bs}_$v.json'));print(d['value'], d['ms_per_step'])")"
  cp "$f" $out/kernel_stats_bs${bs}_head$v.csv
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"  total GPU kernel time per step {tot/23/1e6:.3f} ms")
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("head_stats", "head_finish", "rows_stats", "step_epilogue")) or (int(r["Calls"]) == 23 and "Cijk" in n and 35000 < float(r["AverageNs"]) < 120000):
        print(f'  {r["Calls"]:>5} calls  avg {float(r["AverageNs"])/1e3:8.1f} us   {n[:100]}')
PY
done
done
