mkdir -p gpurun_out/fuzz
( echo "## fuzz_parity.py 6000 91 (final round-4 kernels, head $(cat gpurun_out/fuzz/head 2>/dev/null))"; timeout -k 10 500 python tools/fuzz_parity.py 6000 91 | tail -2 ) > gpurun_out/fuzz/f1.txt 2>&1 &&
( echo "## fuzz_parity.py 800 92 wide"; timeout -k 10 300 python tools/fuzz_parity.py 800 92 wide | tail -1 ) > gpurun_out/fuzz/f2.txt 2>&1 &&
( echo "## fuzz_batch.py 800 93"; timeout -k 10 400 python tools/fuzz_batch.py 800 93 | tail -1 ) > gpurun_out/fuzz/f3.txt 2>&1
cat gpurun_out/fuzz/f*.txt
