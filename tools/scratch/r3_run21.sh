set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_one_launch or epilogue" > gpurun_out/r3/t_f.log 2>&1; tail -25 gpurun_out/r3/t_f.log
timeout 900 python -m pytest tests/test_gpu_attack.py -x -q -k "fused or single_gpu or bench_contract" > gpurun_out/r3/t_fa.log 2>&1; tail -15 gpurun_out/r3/t_fa.log
