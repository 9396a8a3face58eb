set -u
export TMPDIR=/tmp
for cap in 512 256 128 64 384; do
for bs in 64 8 16; do
VAA_K1T_FPWGS=$cap timeout 200 python tools/k1t_bench.py $bs
done; done
