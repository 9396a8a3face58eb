set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "tiles" > gpurun_out/r3/t_tiles.log 2>&1; tail -25 gpurun_out/r3/t_tiles.log
timeout 900 python -m pytest tests/test_gpu_attack.py -x -q -k "two_ranks_vs_single or patch_embed_grad_path or trajectory or reference" > gpurun_out/r3/t_att.log 2>&1; tail -5 gpurun_out/r3/t_att.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err; tail -2 gpurun_out/r3/bench_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_b.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'])
print('roofline',{k:d['roofline'][k] for k in ('frac','mean_us','min_us','samples')})
print(d['hot_path_us_per_step'], d['hot_path_launches_per_step'])
for k,v in d['roofline_kernels'].items(): print(k[:60], v['launches_per_step'], round(v['mean_us'],2), round(v['min_us'],2))
PY
timeout 300 python tools/kbench.py --iters 20 > gpurun_out/r3/kbench_b.json 2>gpurun_out/r3/kbench_b.err; tail -3 gpurun_out/r3/kbench_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/kbench_b.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict) and 'mean_us' in v: print(k, round(v['mean_us'],2))
PY
