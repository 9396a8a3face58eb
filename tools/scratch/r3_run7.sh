set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "embed or epilogue or tiles" > gpurun_out/r3/t_k2e.log 2>&1; tail -8 gpurun_out/r3/t_k2e.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o kb -- python tools/kbench.py --iters 20 > gpurun_out/r3/kbench_d.json 2>gpurun_out/r3/kbench_d.err
cp "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" gpurun_out/r3/kbench_d_kernel_stats.csv
grep "vaa::" gpurun_out/r3/kbench_d_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/kbench_d.json').read().strip().splitlines()[-1])
for k,v in d['suite'].items(): print(k, round(v['mean_us'],2))
PY
for bs in 8 32; do timeout 200 python tools/kbench.py --iters 20 --bs $bs > gpurun_out/r3/kbench_d$bs.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/r3/kbench_d$bs.json').read().strip().splitlines()[-1])
print('bs',$bs,{k: round(v['mean_us'],2) for k,v in d['suite'].items() if k.startswith('K2e') or k.startswith('K1')})
PY
done
