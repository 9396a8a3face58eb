export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "embed or fused or tiles" 2>&1 | tail -2
timeout 300 python tools/k2f_bench.py 64 40 24 8 2>&1 | grep "^B=" | cut -c1-230
timeout 200 python tools/soak_loss.py --seconds 60 --seed 54 2>&1 | tail -1
