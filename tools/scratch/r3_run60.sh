export TMPDIR=/tmp
cp roboticattack_amd/libvaa_hip.so /tmp/tree.so
cp tools/scratch/libs/two.so roboticattack_amd/libvaa_hip.so
for fs in 8 4 16; do
for m in 0 2; do VAA_K1T_FSPLIT=$fs VAA_K1T_ABL=$m timeout 100 python tools/k1t_bench.py 64 2>/dev/null | sed "s/^/fsplit $fs abl $m: /"; done
done
VAA_K1T_ABL=0 timeout 100 python tools/k1t_bench.py 8 2>/dev/null | sed "s/^/B8: /"
VAA_K1T_ABL=0 timeout 100 python tools/k1t_bench.py 16 2>/dev/null | sed "s/^/B16: /"
cp /tmp/tree.so roboticattack_amd/libvaa_hip.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "tiles or k1" 2>&1 | grep "passed\|failed"
