for sp in 1 2 3 4 1 3; do
  TINYORB_BATCH_SPLIT=$sp python bench.py --cpu-sample 0 --no-single-frame --no-host-out --repeats 5 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r=json.loads(ln); print('split $sp: ms_per_step %.4f  repeats %s  kernels %s' % (r['ms_per_step'], ['%.4f'%x for x in r['repeats_ms_per_step']], {k:round(v,4) for k,v in r['roofline']['all_kernels_ms_per_step'].items()}))
"
done
