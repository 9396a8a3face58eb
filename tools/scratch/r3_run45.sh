export TMPDIR=/tmp
timeout 300 python tools/k3_onepass_check.py 2>&1 | grep "^B=\|Error\|error" | head
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_pass" 2>&1 | grep -v "^$" | tail -30
