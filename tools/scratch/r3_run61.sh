export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "under_capture or one_pass" 2>&1 | grep -v "^$" | tail -15
