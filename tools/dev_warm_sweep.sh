#!/bin/bash
# does a short timed region (the driver's --steps 20 --warmup 5) depend on how warm the GPU is?
for w in 5 5 200 2000 5 2000; do
    python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('warmup',$w,'value %.1fM'%(o['value']/1e6),'ms/step %.4f'%o['ms_per_step'],'frac %.3f'%r['frac'],'k_ms %.3f'%r['kernel_ms_avg'])"
done
