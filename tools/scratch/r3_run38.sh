export TMPDIR=/tmp
timeout 300 python tools/cold_probe.py 64 2>&1 | grep "^B="
timeout 300 python tools/cold_probe.py 8 2>&1 | grep "^B="
