export TMPDIR=/tmp
for b in 64 48 24; do timeout 300 python tools/cold_probe.py $b 2>&1 | grep "^B=" | grep -v "GEMMs"; done
timeout 300 python tools/k2f_bench.py 64 48 2>&1 | grep "^B="
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "embed or soak or fused" 2>&1 | grep "passed\|failed"
