#!/usr/bin/env bash
# hipBLASLt / rocBLAS selections for the GEMM shapes of the per-rank batch sizes given (TunableOp tuning pass; run on the MI355X box):
#   gpurun --timeout 2400 -- 'bash tools/tune_gemms.sh 8 4'
# Starts from the shipped selections (already-tuned shapes are kept and skipped) and leaves the union in gpurun_out/tune/all_0.csv;
# copy that file to roboticattack_amd/tunableop/openvla7b_mi355x<ordinal>.csv for ordinals 0..7.
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p gpurun_out/tune
cp roboticattack_amd/tunableop/openvla7b_mi355x0.csv gpurun_out/tune/all_0.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0
export PYTORCH_TUNABLEOP_FILENAME="gpurun_out/tune/all_.csv"
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="${TUNE_MS:-12}" PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2
for bs in "$@"; do
  t0=$(date +%s)
  python bench.py --bs "${bs}" --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-suite > "gpurun_out/tune/bench_bs${bs}.json" 2> "gpurun_out/tune/bench_bs${bs}.err"
  echo "bs=${bs}: rc=$? $(( $(date +%s) - t0 )) s, $(wc -l < gpurun_out/tune/all_0.csv) lines"
done
