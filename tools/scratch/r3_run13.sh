set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "epilogue or tiles or loss" > gpurun_out/r3/t_epi.log 2>&1; tail -12 gpurun_out/r3/t_epi.log
timeout 1500 python -m pytest tests/test_gpu_attack.py -x -q > gpurun_out/r3/t_att.log 2>&1; tail -12 gpurun_out/r3/t_att.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite > gpurun_out/r3/bench_f.json 2> gpurun_out/r3/bench_f.err; tail -2 gpurun_out/r3/bench_f.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_f.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'])
print('roofline',{k:d['roofline'][k] for k in ('frac','mean_us','min_us','samples')})
print(d['hot_path_us_per_step'], d['hot_path_launches_per_step'])
for k,v in d['roofline_kernels'].items(): print(k[:60], v['launches_per_step'], round(v['mean_us'],2), round(v['min_us'],2))
print({k:(round(v['ms_per_step'],2), round(v['images_per_s'],1)) for k,v in d['per_rank_step'].items() if k.startswith('bs')})
PY
