set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o pc -- python tools/prof_check.py > gpurun_out/r3/prof_check.json 2> gpurun_out/r3/prof_check.err
cp "$(find /tmp/pc -name '*kernel_stats.csv' | head -1)" gpurun_out/r3/prof_check_kernel_stats.csv
timeout 200 python tools/prof_check.py > gpurun_out/r3/prof_check_noprof.json 2>> gpurun_out/r3/prof_check.err
bash tools/tune_gemms.sh 8 4 > gpurun_out/r3/tune.log 2>&1
for bs in 8 4; do
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tune/all_.csv timeout 300 python bench.py --bs $bs --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-suite > gpurun_out/r3/b${bs}_tuned.json 2> gpurun_out/r3/b${bs}_tuned.err
done
cat gpurun_out/r3/tune.log
