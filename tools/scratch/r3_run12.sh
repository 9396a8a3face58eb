set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 120 python tools/k3_onepass_check.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "loss" > gpurun_out/r3/t_loss.log 2>&1; tail -3 gpurun_out/r3/t_loss.log
timeout 300 python tools/soak_loss.py --seconds 60 --seed 5 2>&1 | tail -2
