#!/usr/bin/env bash
# round 4, first GPU pass: the GPU suite, the driver's bench command, then a full-size 8-rank functional run on the one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4a; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4a/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['hot_path_us_per_step'], {k:v for k,v in d['config'].items() if 'bs' in k or 'projected' in k})
print({k:(v['mean_us'],v['launches_per_step']) for k,v in d['roofline_kernels'].items()})
PY
VAA_NO_TN_DGRAD=1 timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 --regions strong --no-cpu-baseline --no-kernel-suite > $out/bench_8ranks_one_gpu.json 2> $out/bench_8ranks.err; echo "8rank rc=$?"
tail -c 1500 $out/bench_8ranks.err
head -c 3000 $out/bench_8ranks_one_gpu.json
