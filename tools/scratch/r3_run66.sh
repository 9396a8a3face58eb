export TMPDIR=/tmp
bash tools/tune_gemms.sh 8 4 16 32 2>&1 | tail -5
cp gpurun_out/tune/all_0.csv gpurun_out/r3/tuned_qkv.csv
diff <(sort roboticattack_amd/tunableop/openvla7b_mi355x0.csv) <(sort gpurun_out/tune/all_0.csv) | grep "^>" | head -20
for i in 0 1 2 3 4 5 6 7; do cp gpurun_out/tune/all_0.csv roboticattack_amd/tunableop/openvla7b_mi355x$i.csv; done
for bs in 8 4 16 32; do
for q in 0 1; do
VAA_FUSED_QKV=$q timeout 600 python bench.py --bs $bs --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-suite --no-per-rank > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
python -c "import json; d=json.load(open('/tmp/o.json')); print('QKV bs $bs fused $q: ms/step %.2f  img/s %.1f  entries %s' % (d['ms_per_step'], $bs*1000/d['ms_per_step'], d['config']['tunableop_entries_loaded']))"
done; done
