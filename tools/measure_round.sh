#!/usr/bin/env bash
# One measurement pass on the MI355X box (run through gpurun from the repo root): kernel-only rocprofv3 stats, PMC traffic, the bench
# under rocprofv3, the bench line without and with the CPU baseline. Outputs land in gpurun_out/measure/; copy what is to be judged into
# profiles/ (see profiles/README.md).
#   gpurun --timeout 1500 -- 'bash tools/measure_round.sh'
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="${root}/gpurun_out/measure"
rm -rf "${out}"; mkdir -p "${out}"
export TMPDIR=/tmp
cd "${root}"
stats_csv() { find "$1" -name '*kernel_stats.csv' | head -1; }

timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -o kb -- python tools/kbench.py --iters 20 > "${out}/kbench.json" 2> "${out}/kbench.err"
cp "$(stats_csv /tmp/prof_kb)" "${out}/kbench_kernel_stats.csv" 2>/dev/null

timeout 400 python tools/pmc_traffic.py > "${out}/traffic.log" 2>&1 && cp gpurun_out/traffic.json "${out}/traffic.json" && cp gpurun_out/traffic.json profiles/traffic_r02.json

timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-suite > "${out}/bench_under_rocprof.json" 2> "${out}/bench_under_rocprof.err"
cp "$(stats_csv /tmp/prof_b)" "${out}/bench_kernel_stats.csv" 2>/dev/null

timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > "${out}/bench_no_cpu.json" 2> "${out}/bench_no_cpu.err"
timeout 900 python bench.py > "${out}/bench.json" 2> "${out}/bench.err"
ls -la "${out}"
