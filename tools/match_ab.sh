timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "match_consecutive or match_frames" 2>&1 | tail -15
rc=${PIPESTATUS[0]}
if [ $rc -ne 0 ]; then exit $rc; fi
TINYORB_MATCH_VALU=1 timeout -k 10 200 python tools/match_rate.py 2>&1 | tail -1
timeout -k 10 200 python tools/match_rate.py 2>&1 | tail -1
