# usage (GPU box): bash tools/collect_intended.sh <tag>  -- the opt-in intended mode: bench line, kernel stats, HBM traffic counters
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r02}; OUT=gpurun_out/${TAG}_intended; mkdir -p $OUT
ARGS="python3 bench.py --mode intended --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --mode intended --steps 20 --warmup 3 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $ARGS > $OUT/write.json 2> $OUT/write.err
python3 bench.py --mode intended > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | cut -c1-400
