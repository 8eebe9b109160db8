for m in 0 1 2 3; do
  echo "== prio $m"
  SFSN_SCAN_PRIO=$m python bench.py --no-cpu-baseline --steps 30 --warmup 6 --time-all 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value',d['value'],'ms',d['ms_per_step'],'single',d['config']['single_stream'], 'sb launch', d['roofline']['launch_ms'], {k:round(v,3) for k,v in d['roofline'].get('other_kernels_ms',{}).items() if 'scan' in k})
"
done
