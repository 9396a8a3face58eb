set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "embed or epilogue or tiles" > gpurun_out/r3/t_k2e.log 2>&1; tail -3 gpurun_out/r3/t_k2e.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o kb -- python tools/kbench.py --iters 20 > gpurun_out/r3/kbench_d.json 2>gpurun_out/r3/kbench_d.err
cp "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" gpurun_out/r3/kbench_d_kernel_stats.csv
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3/kbench_d_kernel_stats.csv')):
    if 'vaa::' in r['Name']: print(r['Name'][:85].ljust(86), r['Calls'].rjust(5), round(float(r['AverageNs'])/1e3,2), round(float(r['MinNs'])/1e3,2))
PY
for bs in 64 8 32; do timeout 200 python tools/kbench.py --iters 20 --bs $bs > gpurun_out/r3/kbench_d$bs.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/r3/kbench_d$bs.json').read().strip().splitlines()[-1])
print('bs',$bs,{k: round(v['mean_us'],2) for k,v in d['suite'].items() if k.startswith('K2e') or k.startswith('K1')})
PY
done
timeout 300 python tools/soak_parity.py --seconds 60 --seed 77 2>&1 | tail -3
