# usage (GPU box): bash tools/match_ab.sh  -- the matcher tests (all three forms), then tools/match_rate.py on the vector unit, int8 MFMA and fp4 MFMA (default)
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "match_consecutive or match_frames" 2>&1 | tail -4
rc=${PIPESTATUS[0]}
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python tools/fuzz_match.py ${1:-100} 21 2>&1 | tail -2
TINYORB_MATCH_VALU=1 timeout -k 10 200 python tools/match_rate.py 2>&1 | tail -1
TINYORB_MATCH_I8=1 timeout -k 10 200 python tools/match_rate.py 2>&1 | tail -1
timeout -k 10 200 python tools/match_rate.py 2>&1 | tail -1
