# Round 4: single 1080p frame (BASELINE config 2 as literally stated) under schedule settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1  # the settings below are tuning switches (ignored without it)
O=gpurun_out/r4; mkdir -p $O
run() { name="$1"; shift; echo -n "$name: "; env "$@" python scripts/single_frame_latency.py 2>&1 | grep "single 1080p" | sed 's/single 1080p frame: //' | cut -c1-200; }
run default X=1
run v3 PIGO_SCAN_VARIANT=3
run v3_nh4 PIGO_SCAN_VARIANT=3 PIGO_NH_GLB=4
run v3_nh4_min128 PIGO_SCAN_VARIANT=3 PIGO_NH_GLB=4 PIGO_REG_MIN_REGIONS=128
run v3_nh4_min512 PIGO_SCAN_VARIANT=3 PIGO_NH_GLB=4 PIGO_REG_MIN_REGIONS=512
run v3_nh4_res0 PIGO_SCAN_VARIANT=3 PIGO_NH_GLB=4 PIGO_REG_RESERVE0_KB=0
run v3_nh9 PIGO_SCAN_VARIANT=3 PIGO_NH_GLB=9
run v3_nh4_s1_100 PIGO_SCAN_VARIANT=3 PIGO_NH_GLB=4 PIGO_REG_S1=100
run v3_nh4_s1_75 PIGO_SCAN_VARIANT=3 PIGO_NH_GLB=4 PIGO_REG_S1=75
