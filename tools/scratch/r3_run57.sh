export TMPDIR=/tmp
cp roboticattack_amd/libvaa_hip.so /tmp/tree.so
for rep in 1 2; do
for v in tree three; do
[ $v = tree ] && cp /tmp/tree.so roboticattack_amd/libvaa_hip.so || cp tools/scratch/libs/$v.so roboticattack_amd/libvaa_hip.so
for b in 64 48; do timeout 300 python tools/cold_probe.py $b 2>&1 | grep "^B=" | grep -v "GEMMs" | sed "s/^/$v /"; done
done; done
cp tools/scratch/libs/three.so roboticattack_amd/libvaa_hip.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "embed" 2>&1 | grep "passed\|failed"
cp /tmp/tree.so roboticattack_amd/libvaa_hip.so
