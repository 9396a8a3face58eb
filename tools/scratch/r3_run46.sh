set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all.log 2>&1; tail -6 gpurun_out/r3/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/k3_onepass_check.py 2>&1 | grep "^B=" | tee gpurun_out/r3/k3_onepass.txt
timeout 400 python tools/kbench.py --iters 20 > gpurun_out/r3/kbench.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/kbench.json'))['suite']
for k in ('K3_full_rows_fwd_bwd','K3_full_two_launches','K3_loss_rows_fwd_bwd'): print(k, round(d[k]['mean_us'],2), round(d[k]['min_us'],2))
PY
timeout 600 python tools/pmc_traffic.py > gpurun_out/r3/traffic.log 2>&1; cp gpurun_out/traffic.json gpurun_out/r3/traffic.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/traffic.json'))
for k,v in d['ops'].items(): print(k, round(v['hbm_bytes_per_launch']/1e6,2),'MB')
PY
