# usage (GPU box): bash tools/ab.sh "ENV1=.. ENV2=.." ...   -- one bench.py run per argument (environment assignments, "-" for none), same box
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  env $cfg python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-host-out --no-single-frame $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_ms_per_step']; print('%-40s' % '$cfg', 'ms/step %.4f' % d['ms_per_step'], 'min %.4f' % d['min_ms_per_step'], {a: round(b,4) for a,b in k.items()})"
done
