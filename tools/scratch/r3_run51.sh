export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k3 or one_pass or loss" 2>&1 | grep "passed\|failed"
