# usage (GPU box): bash tools/abl.sh  -- k_front phase ablation (TINYORB_PHASE_MASK: bit0 B1, bit1 B2, bit2 C0, bit3 C; A always runs)
# and an occupancy experiment (TINYORB_LDS_PAD: extra dynamic LDS -> one workgroup per CU), all in one call on one box.
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --repeats 3 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_ms_per_step']; print('$1', 'ms/step %.4f' % d['ms_per_step'], {a: round(b,4) for a,b in k.items()})"; }
for m in 15 0 1 3 7 8 9 11; do TINYORB_PHASE_MASK=$m run mask=$m; done
TINYORB_LDS_PAD=12000 run lds_pad_1wg_per_cu
TINYORB_NO_SWIZZLE=1 run no_xcd_swizzle
