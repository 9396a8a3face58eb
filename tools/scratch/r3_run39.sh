export TMPDIR=/tmp
mkdir -p gpurun_out/r3
cd /tmp
for bs in 8 4; do
  rm -rf /tmp/prof_t
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python /root/repo/bench.py --bs $bs --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank > /tmp/b$bs.json 2>/tmp/b$bs.err
  f=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
  python -c "import json; d=json.load(open('/tmp/b$bs.json')); print('bs $bs under rocprof ms/step', d['ms_per_step'], 'host_enqueue', d['host_enqueue_ms_per_step'])"
  python /root/repo/tools/gpu_idle.py $f 3 | sed "s/^/bs $bs: /"
  VAA_TOWER_STREAMS=0 timeout 600 python /root/repo/bench.py --bs $bs --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank > /tmp/c$bs.json 2>/tmp/c$bs.err
  python -c "import json; d=json.load(open('/tmp/c$bs.json')); print('bs $bs one stream, unprofiled ms/step', d['ms_per_step'], 'host_enqueue', d['host_enqueue_ms_per_step'])"
  timeout 600 python /root/repo/bench.py --bs $bs --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-suite --no-per-rank > /tmp/d$bs.json 2>/tmp/d$bs.err
  python -c "import json; d=json.load(open('/tmp/d$bs.json')); print('bs $bs two streams, unprofiled ms/step', d['ms_per_step'], 'host_enqueue', d['host_enqueue_ms_per_step'])"
done
