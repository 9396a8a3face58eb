export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 300 python tools/cold_probe.py 64 2>&1 | grep "^B="
( time timeout 1500 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/bench_default.json'))
print(d['value'], d['ms_per_step'], d['wall_s'])
PY
