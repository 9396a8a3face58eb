set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
for bs in 8 4; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$bs -o b$bs -- python bench.py --bs $bs --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite > gpurun_out/r3/b${bs}_prof.json 2> gpurun_out/r3/b${bs}_prof.err
  cp "$(find /tmp/p$bs -name '*kernel_stats.csv' | head -1)" gpurun_out/r3/b${bs}_kernel_stats.csv
  timeout 300 python bench.py --bs $bs --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-suite > gpurun_out/r3/b${bs}.json 2> gpurun_out/r3/b${bs}.err
done
timeout 300 python bench.py --bs 64 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite > gpurun_out/r3/b64.json 2> gpurun_out/r3/b64.err
ls -la gpurun_out/r3
