set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all.log 2>&1; grep "passed\|failed" gpurun_out/r3/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/measure_round.sh > gpurun_out/r3/measure_inner.log 2>&1
timeout 300 python tools/k2f_bench.py 64 40 24 8 2>&1 | grep "^B=" > gpurun_out/measure/k2_fused.txt
