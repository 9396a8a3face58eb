set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_model_ops.py -x -q -k "two_streams or fused_ops" > gpurun_out/r3/t_model.log 2>&1; tail -3 gpurun_out/r3/t_model.log
timeout 1200 python -m pytest tests/test_gpu_attack.py -x -q -k "bench" > gpurun_out/r3/t_bench.log 2>&1; tail -5 gpurun_out/r3/t_bench.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "loss" > gpurun_out/r3/t_loss.log 2>&1; tail -3 gpurun_out/r3/t_loss.log
run() { # tag bs streams tunedfile
  if [ -n "$4" ]; then export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$4; else unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_TUNING PYTORCH_TUNABLEOP_FILENAME; fi
  VAA_TOWER_STREAMS=$3 timeout 300 python bench.py --bs $2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-suite --no-per-rank > gpurun_out/r3/ab_$1.json 2> gpurun_out/r3/ab_$1.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3/ab_$1.json').read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],2), 'ms', round(d['config']['images_per_s'],1), 'img/s tun', d['config']['tunableop_entries_loaded'])
PY
}
for rep in 1 2; do
run b8_s0_def_$rep 8 0 ""
run b8_s1_def_$rep 8 1 ""
run b8_s0_tun_$rep 8 0 gpurun_out/tune/all_.csv
run b8_s1_tun_$rep 8 1 gpurun_out/tune/all_.csv
run b4_s0_def_$rep 4 0 ""
run b4_s1_def_$rep 4 1 ""
run b4_s0_tun_$rep 4 0 gpurun_out/tune/all_.csv
run b4_s1_tun_$rep 4 1 gpurun_out/tune/all_.csv
done
run b64_s0_def 64 0 ""
run b64_s1_def 64 1 ""
