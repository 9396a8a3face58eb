export TMPDIR=/tmp
for rep in 1 2 3; do
for bs in 4 8; do
for ts in 0 1; do
  VAA_TOWER_STREAMS=$ts timeout 600 python bench.py --bs $bs --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-suite --no-per-rank > /tmp/o.json 2>/tmp/o.err
  python -c "import json; d=json.load(open('/tmp/o.json')); print('AB bs $bs streams $ts rep $rep: ms/step %.2f host_enqueue %.2f host_cpu %.2f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['host_cpu_ms_per_step']))"
done; done; done
