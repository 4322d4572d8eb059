# usage (GPU box): bash tools/ab_old_new.sh name1 name2 ...  -- same-box A/B of whole libraries tinyslam_amd/libtinyorb_<name>.so (built here from
# other commits' sources), three rounds, alternating; the tree's own library is put back at the end.  An A/B through a run-time switch inside ONE
# binary does not see what a change costs the binary as a whole (registers, code around the switch) -- round 4 learnt that the hard way.
out=gpurun_out/abso; mkdir -p $out
export TINYORB_ALLOW_ABI=1  # libraries of older commits report an older ABI version; the bench uses nothing that changed
cp tinyslam_amd/libtinyorb.so $out/keep.so
run() { python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-host-out --no-single-frame > $out/$1.json 2> $out/$1.err || { echo fail $1; tail -3 $out/$1.err; cp $out/keep.so tinyslam_amd/libtinyorb.so; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
print("$1:", round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
}
for i in 1 2 3; do
  for n in "$@"; do cp tinyslam_amd/libtinyorb_$n.so tinyslam_amd/libtinyorb.so; run $n$i; done
done
cp $out/keep.so tinyslam_amd/libtinyorb.so
