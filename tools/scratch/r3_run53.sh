export TMPDIR=/tmp
for b in 64 40 32 16; do timeout 300 python tools/cold_probe.py $b 2>&1 | grep "^B=" | grep -v "GEMMs"; done
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "embed or soak" 2>&1 | grep "passed\|failed"
