run() { python bench.py --steps 3 --repeats 1 --preheat-ms 50 --cpu-sample 0 --no-host-out 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r=json.loads(ln); print('$1 ->', round(r['single_frame_us'],1), round(r['single_frame_loop_fps']), round(r['single_frame_loop_pinned_fps']))
"; }
run default
GPU_MAX_HW_QUEUES=8 run hwq8
GPU_MAX_HW_QUEUES=2 run hwq2
HSA_ENABLE_SDMA=0 run nosdma
run default
