export TMPDIR=/tmp
mkdir -p gpurun_out/r3
cd /tmp
for q in 0 1; do
rm -rf /tmp/prof_q$q
VAA_FUSED_QKV=$q timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q$q -o p -- python /root/repo/bench.py --bs 8 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank > /tmp/o$q.json 2>/tmp/o$q.err
cp $(find /tmp/prof_q$q -name '*kernel_stats.csv' | head -1) /root/repo/gpurun_out/r3/bs8_qkv$q.csv
done
