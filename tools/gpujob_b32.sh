for b in 32 256; do
python bench.py --batch $b --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); P=$b
print('batch',P,'value',round(d['value'],1),'ms/step',round(d['ms_per_step'],3),'kernel_sum_ms',round(sum(d['top_kernels_ms'].values())/d['steps'],3))
print({k: round(v/d['steps']/P*1000,1) for k,v in d['top_kernels_ms'].items()}, 'us per proof')"
done
