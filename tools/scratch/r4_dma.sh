#!/usr/bin/env bash
# head_stats_kernel (LDS-DMA form): parity tests, back-to-back / cold times
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"
out=$root/gpurun_out/r4d; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "head_loss" > $out/pytest_head.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_head.log
timeout 600 python tools/head_bench.py ${ROWS:-128 96 64 32 16 2} 2>&1 | tee $out/head_bench.txt | grep -E "^R=|COLD|in a stream"
