#!/usr/bin/env bash
# Round 5: BASELINE configs 2 / 4 / 5 end to end under rocprofv3 (the product loops' own inner_step through `bench.py --attack`), config 5's
# backward A/B (K2'-MULTI against the ViT patch-embed conv backward in torch + K2-MULTI), and the in-step A/B of the fused LM head at bs=64.
#   gpurun --timeout 2400 -- 'bash tools/measure_configs.sh [cfg] [cfg5ab] [head] [k3s] [k3scols]'      (default: cfg cfg5ab head)
# Outputs: gpurun_out/cfg/{cfgN_kernel_stats.csv, cfgN.json, cfgN_summary.txt, ...}; copy what is to be judged into profiles/r05_*.
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="${root}/gpurun_out/cfg"
mkdir -p "${out}"
export TMPDIR=/tmp
cd "${root}"
parts="${*:-cfg cfg5ab head}"
STEPS=20; WARM=3; N=$((STEPS + WARM))   # --profile-steps 0: every kernel of the CSV has N calls per launch site
common="--steps ${STEPS} --warmup ${WARM} --no-cpu-baseline --no-kernel-suite --no-per-rank --no-configs --profile-steps 0"

run() {  # tag, env assignments (may be empty), bench arguments
  local tag="$1" envs="$2"; shift 2
  env ${envs} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_${tag}" -o p -- python bench.py ${common} --full-out "gpurun_out/cfg/${tag}_full.json" "$@" > "${out}/${tag}.json" 2> "${out}/${tag}.err"
  local f; f="$(find "/tmp/prof_${tag}" -name '*kernel_stats.csv' | head -1)"
  cp "${f}" "${out}/${tag}_kernel_stats.csv" 2>/dev/null
  { echo "== ${tag}  ${envs}  bench.py $*"; python -c "import json;d=json.load(open('${out}/${tag}.json'));print('   value', d['value'], 'ms_per_step', d['ms_per_step'], 'hot_path_us', d['hot_path_us_per_step'])";
    python tools/kstats.py "${out}/${tag}_kernel_stats.csv" ${N}; } | tee "${out}/${tag}_summary.txt"
  rm -rf "/tmp/prof_${tag}"
}

for p in ${parts}; do
  case "${p}" in
    cfg)
      run cfg2 "" --attack uada --bs 16 --geometry false
      run cfg4 "" --attack tma --bs 8
      run cfg5 "" --attack upa --bs 4 --resize-patch --patch 3,100,100
      ;;
    cfg5ab)  # the same UPA step with the pixel path: conv backward of both towers in torch + K2-MULTI on the bf16 pixel gradient
      run cfg5_pixel_path "VAA_FUSED_EMBED_GRAD=0" --attack upa --bs 4 --resize-patch --patch 3,100,100
      run cfg5_again "" --attack upa --bs 4 --resize-patch --patch 3,100,100
      ;;
    k3s)     # round 6: K3s on every step + K3h on 1 of 50 (the default) against K3h + finish + the 256-column GEMM on every step, inside the bs=64 step, A B B A
      run k3s_cadence_1 ""
      run k3h_every_1 "VAA_HEAD_EVERY_STEP=1"
      run k3h_every_2 "VAA_HEAD_EVERY_STEP=1"
      run k3s_cadence_2 ""
      ;;
    k3scols) # round 6: K3s with 8 action columns per workgroup (32 workgroups per row block, the default) against 16, inside the bs=64 step, A B B A
      run k3s_cols8_1 ""
      run k3s_cols16_1 "VAA_K3S_COLS=16"
      run k3s_cols16_2 "VAA_K3S_COLS=16"
      run k3s_cols8_2 ""
      ;;
    head)    # fused LM head (K3h) against hipBLASLt head + K3 statistics, inside the bs=64 step, A B B A
      run head_gemm_1 "VAA_FUSED_HEAD=0"
      run head_fused_1 "VAA_FUSED_HEAD=1"
      run head_fused_2 "VAA_FUSED_HEAD=1"
      run head_gemm_2 "VAA_FUSED_HEAD=0"
      ;;
  esac
done
ls -la "${out}"
