#!/usr/bin/env bash
# Same-box A/B of an environment switch by un-profiled step time (A B B A): ms per step of the bs=64 headline step and of the bs=8 per-rank step.
#   gpurun --timeout 1500 -- 'bash tools/ab_env.sh VAA_RES_PREFETCH=0 VAA_RES_PREFETCH=1'      -> gpurun_out/ab/<A>__<B>.txt
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
A="$1"; B="$2"; out="${root}/gpurun_out/ab"; mkdir -p "${out}"; cd "${root}"; export TMPDIR=/tmp
f="${out}/${A//[^A-Za-z0-9_=]/_}__${B//[^A-Za-z0-9_=]/_}.txt"
common="--no-cpu-baseline --no-kernel-suite --no-per-rank --no-configs --profile-steps 0"
{
  for bs in 64 8; do
    steps=$([ "${bs}" = 64 ] && echo 30 || echo 80)
    for e in "${A}" "${B}" "${B}" "${A}"; do
      env ${e} timeout 600 python bench.py --bs ${bs} --steps ${steps} --warmup 5 ${common} --full-out gpurun_out/ab/full.json 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bs=${bs}  %-28s %9.3f ms/step  %8.4f steps/s' % ('${e}', d['ms_per_step'], d['value']))"
    done
  done
} | tee "${f}"
