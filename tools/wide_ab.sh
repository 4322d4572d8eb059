# usage (GPU box): bash tools/wide_ab.sh  -- literal mode at wide frame sizes, full-width bands against column tiles of several widths
cd $GRAFT_REPO_ROOT
r() { echo "TILE_W=$1 $2"; if [ "$1" = 0 ]; then python3 tools/wide_rate.py $2 2>&1 | grep "fused "; else TINYORB_TILE_W=$1 python3 tools/wide_rate.py $2 2>&1 | grep "fused "; fi; }
r 0 "1280 720 128 2"
r 0 "1920 1080 64 3"; r 960 "1920 1080 64 3"; r 640 "1920 1080 64 3"
r 0 "1440 1080 64 3"; r 720 "1440 1080 64 3"
r 0 "2560 1440 32 3"; r 1280 "2560 1440 32 3"; r 864 "2560 1440 32 3"
r 0 "3840 2160 32 3"; r 960 "3840 2160 32 3"
