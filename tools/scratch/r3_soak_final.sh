export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 1500 python tools/soak_parity.py --seconds 1200 --seed 211 2>&1 | tail -1
timeout 800 python tools/soak_loss.py --seconds 600 --seed 212 2>&1 | tail -1
