export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k3 or one_pass or loss or two_streams" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_attack.py -x -q -k "uada or tma or trajectory or single" 2>&1 | tail -3
