export TMPDIR=/tmp
echo "levels 2"; timeout 300 python tools/k3_onepass_check.py 2>&1 | grep "^B="
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_pass" 2>&1 | tail -2
cp tools/scratch/libs/libvaa_l1.so roboticattack_amd/libvaa_hip.so
echo "levels 1"; timeout 300 python tools/k3_onepass_check.py 2>&1 | grep "^B="
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_pass" 2>&1 | tail -2
