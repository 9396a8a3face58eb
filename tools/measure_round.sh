#!/usr/bin/env bash
# One measurement pass on the MI355X box (run through gpurun from the repo root): kernel-only rocprofv3 stats, PMC traffic, the bench
# under rocprofv3, the bench line without and with the CPU baseline. Outputs land in gpurun_out/measure/; copy what is to be judged into
# profiles/ as rNN_* (see profiles/README.md).
#   gpurun --timeout 2400 -- 'bash tools/measure_round.sh'
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="${root}/gpurun_out/measure"
rm -rf "${out}"; mkdir -p "${out}"
export TMPDIR=/tmp
cd "${root}"
stats_csv() { find "$1" -name '*kernel_stats.csv' | head -1; }

# the suite's own figures come from an un-profiled run (entries timed as back-to-back calls in a stream are host-bound under the profiler);
# the profiled run is for the per-kernel statistics only
timeout 300 python tools/kbench.py --iters 20 > "${out}/kbench.json" 2> "${out}/kbench.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -o kb -- python tools/kbench.py --iters 20 > "${out}/kbench_profiled.json" 2>> "${out}/kbench.err"
cp "$(stats_csv /tmp/prof_kb)" "${out}/kbench_kernel_stats.csv" 2>/dev/null

timeout 400 python tools/pmc_traffic.py > "${out}/traffic.log" 2>&1 && cp gpurun_out/traffic.json "${out}/traffic.json"

# the driver's N=1 command under rocprofv3 (20 steps like the driver's run; the CPU baseline and the stand-alone suite are not GPU work of the step)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank --no-configs > "${out}/bench_under_rocprof.json" 2> "${out}/bench_under_rocprof.err"
cp "$(stats_csv /tmp/prof_b)" "${out}/bench_kernel_stats.csv" 2>/dev/null
# the same command un-profiled: its in-step per-dispatch figures are what the record carries
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank --no-configs > "${out}/bench_same_cmd.json" 2> "${out}/bench_same_cmd.err"

timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --full-out gpurun_out/measure/bench_no_cpu_full.json > "${out}/bench_no_cpu.json" 2> "${out}/bench_no_cpu.err"
timeout 1200 python bench.py --full-out gpurun_out/measure/bench_full.json > "${out}/bench.json" 2> "${out}/bench.err"
# the form the driver runs at round end: ONE compact line on stdout (+ the full record beside it)
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/measure/bench_driver_cmd_full.json > "${out}/bench_driver_cmd.json" 2> "${out}/bench_driver_cmd.err"
# two ranks on the one GPU over gloo: the N > 1 code path at full size, both timed regions (functional evidence, not a scaling number)
timeout 900 python bench.py --gpus 2 --steps 4 --warmup 1 --full-out gpurun_out/measure/bench_2ranks_one_gpu_full.json > "${out}/bench_2ranks_one_gpu.json" 2> "${out}/bench_2ranks_one_gpu.err"
# eight ranks on the one GPU, the strong-scaling region alone (bs = 8 per rank, BASELINE config 3's split): the full-size 8-rank functional run.
# VAA_NO_TN_DGRAD=1 drops the 12.9 GB of resident transposed weights per rank so that eight copies of the model fit the 288 GB
VAA_NO_TN_DGRAD=1 timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 --regions strong --full-out gpurun_out/measure/bench_8ranks_one_gpu_full.json > "${out}/bench_8ranks_one_gpu.json" 2> "${out}/bench_8ranks_one_gpu.err"
timeout 200 python tools/k3_onepass_check.py > "${out}/k3_onepass.txt" 2>&1
# K3s: the slice-only head against K3h + finish + the 256-column GEMM (per dispatch, in a stream, cold), and where its time goes (timing build)
{ timeout 300 python tools/k3s_bench.py 128 64 32 16; echo "== VAA_K3S_COLS=16 (16 action columns per workgroup, 16 workgroups per row block)"; VAA_K3S_COLS=16 timeout 300 python tools/k3s_bench.py 128 16; } > "${out}/k3s_bench.txt" 2>&1
sed -e 's#^out=.*#out="${here}/../libvaa_hip_timing.so"#' -e "s#^here=.*#here=${root}/roboticattack_amd/csrc#" roboticattack_amd/csrc/build.sh > /tmp/build_timing.sh && bash /tmp/build_timing.sh -DVAA_K3S_TIMING > /dev/null 2>&1
{ for c in 8 16; do echo "== ${c} action columns per workgroup, warm then behind a 1 GiB copy"; VAA_K3S_COLS=${c} VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py; VAA_K3S_COLS=${c} VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py cold; done; } > "${out}/k3s_stamps.txt" 2>&1
rm -f roboticattack_amd/libvaa_hip_timing.so
# the validation pass of the data-parallel loop (bs = 8, 12 batches, A B A B against the per-batch read-back), alone and under rocprofv3
timeout 600 python tools/val_bench.py openvla-7b 12 > "${out}/val_bench.json" 2> "${out}/val_bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_val -o v -- python tools/val_bench.py openvla-7b 12 > "${out}/val_bench_under_rocprof.json" 2>> "${out}/val_bench.err"
cp "$(stats_csv /tmp/prof_val)" "${out}/val_kernel_stats.csv" 2>/dev/null
timeout 300 python tools/head_bench.py 128 64 16 > "${out}/head_bench.txt" 2>&1
# the per-rank step of BASELINE config 3 (bs = 8: the LM head runs fused with K3's statistics) under rocprofv3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -o b8 -- python bench.py --bs 8 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank --no-configs > "${out}/bench_bs8_under_rocprof.json" 2> "${out}/bench_bs8_under_rocprof.err"
cp "$(stats_csv /tmp/prof_b8)" "${out}/bench_bs8_kernel_stats.csv" 2>/dev/null
ls -la "${out}"
