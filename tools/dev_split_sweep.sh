for sp in 0 1; do
  for s in 20 400 4000; do
    python bench.py --steps $s --warmup 5 --opt wave_split=$sp --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read()); r=o['roofline']
print('wave_split',$sp,'steps',$s,'value %.1fM'%(o['value']/1e6),'ms/step %.4f'%o['ms_per_step'],'frac %.3f'%r['frac'],'in_flight %.1f'%r['launches_in_flight'],'k_ms %.3f'%r['kernel_ms_avg'])"
  done
done
