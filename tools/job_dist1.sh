# the bench's multi-rank code path on ONE GPU: launched through torch.distributed.run, every collective forced
mkdir -p gpurun_out/dist1
KIEZ_AMD_FORCE_COLLECTIVES=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 5 --warmup 2 --no-others --no-cpu-baseline > gpurun_out/dist1/ns_forced.json 2> gpurun_out/dist1/err.txt
tail -3 gpurun_out/dist1/err.txt
python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/dist1/ns_forced.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','n_gpus','collective_ms_per_step','collective_traffic_per_step')})
print(j['config']['parallelism']); print(j['check'])
PY
for wl in c3 c4s; do
KIEZ_AMD_FORCE_COLLECTIVES=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --workload $wl --steps 3 --warmup 1 --no-others --no-cpu-baseline > gpurun_out/dist1/${wl}_forced.json 2>> gpurun_out/dist1/err.txt
python3 - $wl <<'PY'
import json,sys
j=json.loads(open(f'gpurun_out/dist1/{sys.argv[1]}_forced.json').read().strip().splitlines()[-1])
print(sys.argv[1], {k:j[k] for k in ('value','ms_per_step','collective_ms_per_step','collective_traffic_per_step')}, j['check'])
PY
done
