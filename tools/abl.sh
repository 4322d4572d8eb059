python bench.py --cpu-sample 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['roofline']['all_kernels_ms_per_step'], d['keypoints_per_frame'])"
