run() { python bench.py --steps 4000 --warmup 24 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read()); r=o['roofline']
print('%-40s value %.1fM frac %.3f'%(' '.join(sys.argv[1:]), o['value']/1e6, r['frac']))" "$@"; }
for p in 0 12 14 16 18 20 22; do run --opt wave_prio=$p; done
for d in 6 7 8 10 12; do run --depth $d; done
for q in 8 12 16 24 32; do GPU_MAX_HW_QUEUES=$q run --depth 8; done
for p in 16 18 20; do run --depth 10 --opt wave_prio=$p; done
