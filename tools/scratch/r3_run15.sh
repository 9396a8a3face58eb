set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r3/t_kern.log 2>&1; tail -5 gpurun_out/r3/t_kern.log
timeout 1500 python -m pytest tests/test_gpu_attack.py -x -q > gpurun_out/r3/t_att.log 2>&1; tail -4 gpurun_out/r3/t_att.log
timeout 900 python -m pytest tests/test_gpu_model_ops.py -x -q > gpurun_out/r3/t_mod.log 2>&1; tail -3 gpurun_out/r3/t_mod.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
