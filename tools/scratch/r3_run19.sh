export TMPDIR=/tmp
for bands in 2 4 6 8 12; do for B in 64 8; do
VAA_K2_BANDS=$bands timeout 200 python tools/k2_records_check.py $B 2>/dev/null | grep "product" | sed "s/^/bands $bands B $B /"
done; done
