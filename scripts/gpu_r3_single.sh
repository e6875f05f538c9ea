# Round-3: single 1080p frame (BASELINE config 2 as literally stated) under tile-geometry / hand-over settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
run() { name="$1"; shift; echo -n "$name: "; env "$@" python scripts/single_frame_latency.py 2>&1 | grep "single 1080p" | sed 's/single 1080p frame: //' | cut -c1-200; }
run default X=1
run rules_16_8 "PIGO_TILE_RULES=6,16,16384;6,8,40960"
run rules_8 "PIGO_TILE_RULES=6,8,40960"
run rules_16 "PIGO_TILE_RULES=6,16,40960"
run rules_32_16_8 "PIGO_TILE_RULES=6,32,12288;6,16,24576;6,8,40960"
run gth8 PIGO_GLOBAL_TH=8
run gth8_r16_8 PIGO_GLOBAL_TH=8 "PIGO_TILE_RULES=6,16,16384;6,8,40960"
run v3 PIGO_SCAN_VARIANT=3
run graph PIGO_GRAPH_FRAMES=4
