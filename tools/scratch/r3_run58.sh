export TMPDIR=/tmp
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-suite > /tmp/b.json 2>/tmp/b.err || tail -5 /tmp/b.err
python - <<'PY'
import json
d=json.load(open('/tmp/b.json'))
print(d['value'], d['hot_path_us_per_step'])
for k,v in d['per_rank_step'].items():
    if isinstance(v,dict): print(k, round(v['ms_per_step'],1), round(v['hot_path_us_per_step'],1), {a:round(b,1) for a,b in v['hot_path_ops_us'].items()})
PY
