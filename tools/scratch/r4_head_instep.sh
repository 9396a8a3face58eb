#!/usr/bin/env bash
# in-step A/B of the fused LM head: rocprofv3 kernel stats of the bench step with VAA_FUSED_HEAD=0 / 1 (and the staging forms VAA_HEAD_DMA) (traces stay in /tmp)
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"
out=$root/gpurun_out/r4h; mkdir -p $out
export TMPDIR=/tmp
for bs in ${BSS:-64}; do
for cfg in ${CFGS:-0:0 1:0 1:1 1:3 0:0}; do
  v=${cfg%%:*}; dma=${cfg##*:}
  tag=bs${bs}_head${v}_dma${dma}
  VAA_HEAD_DMA=$dma VAA_FUSED_HEAD=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o h -- python bench.py --bs $bs --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank --profile-steps 0 > $out/bench_$tag.json 2> $out/bench_$tag.err
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  echo "== bs=$bs VAA_FUSED_HEAD=$v VAA_HEAD_DMA=$dma  $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));print(d['value'], d['ms_per_step'])")"
  cp "$f" $out/kernel_stats_$tag.csv
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
  rm -rf /tmp/prof_$tag
done
done
