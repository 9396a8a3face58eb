# second, longer tuning pass of the six fused-q/k/v shapes (TUNE_MS per candidate), then an A/B of the step against the shipped selections
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/tuneq
grep -v "_12288" roboticattack_amd/tunableop/openvla7b_mi355x0.csv > gpurun_out/tuneq/all_0.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0
export PYTORCH_TUNABLEOP_FILENAME="gpurun_out/tuneq/all_.csv"
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=80 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
for bs in 8 4 16; do python bench.py --bs $bs --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-suite --no-per-rank > /dev/null 2>gpurun_out/tuneq/err$bs.txt; done
grep "_12288" gpurun_out/tuneq/all_0.csv
export PYTORCH_TUNABLEOP_TUNING=0
for bs in 8 4; do
python bench.py --bs $bs --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-suite --no-per-rank 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new bs $bs', d['ms_per_step'])"
env -u PYTORCH_TUNABLEOP_ENABLED -u PYTORCH_TUNABLEOP_TUNING -u PYTORCH_TUNABLEOP_FILENAME python bench.py --bs $bs --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-suite --no-per-rank 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('shipped bs $bs', d['ms_per_step'])"
done
