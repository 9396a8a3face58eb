set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all.log 2>&1; tail -6 gpurun_out/r3/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
