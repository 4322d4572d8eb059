# usage (GPU box): bash tools/content_axis.sh <out.txt>  -- bench.py over the content axis, literal and intended
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/r04_content_axis.txt}
: > $OUT
for mode in ${MODES:-literal intended}; do
for c in flat sparse default dense overflow; do
  python3 bench.py --mode $mode --content $c --cpu-sample 0 --no-single-frame --no-host-out --repeats 3 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r=json.loads(ln)
        k={a:round(b,4) for a,b in r['roofline']['all_kernels_ms_per_step'].items()}
        print('%-8s %-8s %9.0f frames/s  %.4f ms/step  %7.0f kp/frame  %s' % ('$mode', '$c', r['value'], r['ms_per_step'], r['keypoints_per_frame'], k))
" >> $OUT
done
done
cat $OUT
