export TMPDIR=/tmp
timeout 300 python tools/cold_probe.py 64 2>&1 | grep "^B="
timeout 300 python tools/cold_probe.py 8 2>&1 | grep "^B="
timeout 300 python tools/cold_probe.py 24 2>&1 | grep "^B="
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "embed or epilogue or soak" 2>&1 | tail -3
