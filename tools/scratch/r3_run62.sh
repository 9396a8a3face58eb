export TMPDIR=/tmp
timeout 800 python tools/soak_parity.py --seconds 600 --seed 101 2>&1 | tail -2
