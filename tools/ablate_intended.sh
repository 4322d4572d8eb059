# usage (GPU box): bash tools/ablate_intended.sh   -- k_front_i of the intended mode with its phases switched on one by one
# (TINYORB_PHASE_MASK bits: 1 B1, 2 S1, 4 S2, 8 S3+S4, 16 C0 mip, 32 G Gaussian); results are wrong unless the mask is 63
cd $GRAFT_REPO_ROOT
for m in 0 16 48 49 51 55 63; do
  TINYORB_PHASE_MASK=$m python3 bench.py --mode intended --steps 10 --warmup 2 --repeats 2 --cpu-sample 0 --no-host-out --no-single-frame 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_ms_per_step']; print('mask %2d' % $m, 'ms/step %.4f' % d['ms_per_step'], {a: round(b,4) for a,b in k.items()})"
done
