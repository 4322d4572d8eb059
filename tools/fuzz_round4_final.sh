# usage (GPU box): bash tools/fuzz_round4_final.sh [n_single n_batch seed0]  -- the fuzz sweeps at a round's final kernels (round 5: half of the plain RGBA cases under a random fp_contract); output under gpurun_out/fuzz/
N1=${1:-6000}; N2=${2:-800}; S=${3:-91}
mkdir -p gpurun_out/fuzz
( echo "## fuzz_parity.py $N1 $S"; timeout -k 10 1000 python tools/fuzz_parity.py $N1 $S | tail -2 ) > gpurun_out/fuzz/f1.txt 2>&1 &&
( echo "## fuzz_parity.py 800 $((S+1)) wide"; timeout -k 10 300 python tools/fuzz_parity.py 800 $((S+1)) wide | tail -1 ) > gpurun_out/fuzz/f2.txt 2>&1 &&
( echo "## fuzz_parity.py 300 $((S+3)) xwide"; timeout -k 10 300 python tools/fuzz_parity.py 300 $((S+3)) xwide | tail -1 ) > gpurun_out/fuzz/f4.txt 2>&1 &&
( echo "## fuzz_batch.py $N2 $((S+2))"; timeout -k 10 800 python tools/fuzz_batch.py $N2 $((S+2)) | tail -1 ) > gpurun_out/fuzz/f3.txt 2>&1
cat gpurun_out/fuzz/f*.txt
