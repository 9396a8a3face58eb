export TMPDIR=/tmp
cp roboticattack_amd/libvaa_hip.so /tmp/tree.so
for v in g1_early g1; do
cp tools/scratch/libs/$v.so roboticattack_amd/libvaa_hip.so
for b in 64 24; do timeout 300 python tools/cold_probe.py $b 2>&1 | grep "^B=" | grep -v "GEMMs" | sed "s/^/$v /"; done
done
cp /tmp/tree.so roboticattack_amd/libvaa_hip.so
