#!/bin/bash
# depth sweep of the driver's short command (warm order: checks + single-launch measurement first)
for d in 4 5 6 8 10 5 8 10 20; do
    python bench.py --steps 20 --warmup 5 --depth $d --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('depth',$d,'value %.1fM'%(o['value']/1e6),'ms/step %.4f'%o['ms_per_step'],'frac %.3f'%r['frac'],'h2h %.1fM real %.1fM'%(o['host_to_host']['value']/1e6,o['real_data']['value']/1e6))"
done
