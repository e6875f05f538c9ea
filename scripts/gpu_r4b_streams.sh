# the side stream's priority: bench.py's default line (its 1,024-frame and config legs are created next to other plans' streams)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
for spec in "prio1:PIGO_X=1" "prio0:PIGO_SIDE_PRIO=0"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  env PIGO_TUNING=1 $envs timeout 600 python bench.py --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  python -c "
import json;d=json.load(open('$O/bench_$name.json'))
print('$name', d['value'], d['ms_per_step'], 'shard', d['config3_shard']['mwindows_per_s'], 'rot', d['config4_rotated']['upright_faces']['mwindows_per_s'], d['config4_rotated']['rotated_faces']['mwindows_per_s'], '4k', d['config5_4k']['mwindows_per_s'], 'single', d['single_frame']['hbm_resident_ms'], d['single_frame']['host_buffer_ms'], 'ref', d['reference_benchmark']['gpu_ms_per_op'])"
done
export PIGO_TUNING=1
timeout 300 python scripts/ab_r4b.py --frames 1024 --steps 3 --reps 2 "prio1:" "prio0:PIGO_SIDE_PRIO=0" 2>$O/ab_1024.err | tee $O/ab_1024.txt
timeout 300 python -m pytest tests -m gpu -q -x -k "benchmarked_path or big_scales_side_chain or reentrant or batch_api" 2>&1 | tail -2
