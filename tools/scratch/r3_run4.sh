set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "updated_pixels or soak_leg" > gpurun_out/r3/t_new.log 2>&1; tail -12 gpurun_out/r3/t_new.log
timeout 900 python -m pytest tests/test_gpu_attack.py -x -q -k "two_ranks_vs_single" > gpurun_out/r3/t_ddp.log 2>&1; tail -5 gpurun_out/r3/t_ddp.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err; tail -2 gpurun_out/r3/bench_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_a.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'tun',d['config']['tunableop_entries_loaded'])
print('roofline',{k:d['roofline'][k] for k in ('frac','mean_us','min_us','samples')})
print('per_rank',json.dumps(d['per_rank_step'],indent=1))
print('hot', json.dumps(d['hot_path_ops'],indent=1)); print(d['hot_path_us_per_step'], d['hot_path_launches_per_step'])
for k,v in d['roofline_kernels'].items(): print(k[:60], v['launches_per_step'], round(v['mean_us'],2), round(v['min_us'],2))
PY
