cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --force-dist --frames 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/o1.json 2> gpurun_out/o1.err; echo "rc=$? lines=$(wc -l < gpurun_out/o1.json)"; python -c "import json; d=json.load(open('gpurun_out/o1.json')); print(d['value'], d['n_gpus'], d['single_frame'] is not None)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --frames 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/o2.json 2> gpurun_out/o2.err; echo "rc=$? lines=$(wc -l < gpurun_out/o2.json)"; python -c "import json; d=json.load(open('gpurun_out/o2.json')); print(d['value'], d['n_gpus'])"
grep -c "RCCL version" gpurun_out/o1.err gpurun_out/o2.err
