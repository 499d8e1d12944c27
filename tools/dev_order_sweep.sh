run() { python bench.py --warmup 8 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read()); r=o['roofline']
print('%-50s value %.1fM ms/step %.4f frac %.3f'%(' '.join(sys.argv[1:]), o['value']/1e6, o['ms_per_step'], r['frac']))" "$@"; }
for rep in 1 2; do
for o in 0 1 2; do run --steps 4000 --opt wave_order=$o; run --steps 20 --opt wave_order=$o; done
done
