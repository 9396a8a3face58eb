export TMPDIR=/tmp
for rep in 1 2; do
echo "v5 (tree)"; timeout 300 python tools/k3_onepass_check.py 2>&1 | grep "^B="
cp roboticattack_amd/libvaa_hip.so /tmp/tree.so
cp tools/scratch/libs/libvaa_v7.so roboticattack_amd/libvaa_hip.so
echo "v7"; timeout 300 python tools/k3_onepass_check.py 2>&1 | grep "^B="
cp /tmp/tree.so roboticattack_amd/libvaa_hip.so
done
