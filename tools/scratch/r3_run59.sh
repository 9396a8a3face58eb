export TMPDIR=/tmp
cp tools/scratch/libs/abl.so roboticattack_amd/libvaa_hip.so
for b in 64 8; do
for m in 0 1 2; do VAA_K1T_ABL=$m timeout 100 python tools/k1t_bench.py $b 2>/dev/null | sed "s/^/abl $m: /"; done
done
