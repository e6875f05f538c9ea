run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], d['kernel_ms'])"; }
run base X=1
run r_16k32 PIGO_TILE_RULES="6,32,16384;6,16,40960"
run r_24k32 PIGO_TILE_RULES="6,32,24576;6,16,40960"
run r_40k32 PIGO_TILE_RULES="6,32,40960;6,16,49152"
run r_32k_16_48k PIGO_TILE_RULES="6,32,32768;6,16,49152"
run r_32k_16_32k PIGO_TILE_RULES="6,32,32768;6,16,32768"
run r_32k_only PIGO_TILE_RULES="6,32,32768"
run r_32k_16_40_8_56 PIGO_TILE_RULES="6,32,32768;6,16,40960;6,8,57344"
run late2 PIGO_LATE_WAVES=2
run late3 PIGO_LATE_WAVES=3
run nh38 PIGO_NH_LDS=38 PIGO_NH_GLB=38
run glbth8 PIGO_GLOBAL_TH=8
run glbth32 PIGO_GLOBAL_TH=32
