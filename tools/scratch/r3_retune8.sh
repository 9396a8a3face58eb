set -u
mkdir -p gpurun_out/tune8
python - <<PY
src = open("roboticattack_amd/tunableop/openvla7b_mi355x0.csv").read().splitlines()
keep = [l for l in src if not (l.startswith("Gemm") and l.split(",")[1].split("_")[2] in ("2400", "2088", "2048", "16"))]
open("gpurun_out/tune8/all_0.csv", "w").write("\\n".join(keep) + "\\n")
PY
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0
export PYTORCH_TUNABLEOP_FILENAME="gpurun_out/tune8/all_.csv"
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
python bench.py --bs 8 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-suite --no-per-rank > gpurun_out/tune8/tune.json 2> gpurun_out/tune8/tune.err
wc -l gpurun_out/tune8/all_0.csv
export PYTORCH_TUNABLEOP_TUNING=0
for rep in 1 2; do
python bench.py --bs 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-suite --no-per-rank > gpurun_out/tune8/b8_new_$rep.json 2>/dev/null
env -u PYTORCH_TUNABLEOP_ENABLED -u PYTORCH_TUNABLEOP_TUNING -u PYTORCH_TUNABLEOP_FILENAME python bench.py --bs 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-suite --no-per-rank > gpurun_out/tune8/b8_old_$rep.json 2>/dev/null
done
