for N in 4 8; do
  FRESCO_BENCH_BACKEND=gloo FRESCO_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 2 --warmup 1 > gpurun_out/bench_gloo${N}.json 2> gpurun_out/bench_gloo${N}.err
  echo "== N=$N rc=$?"; python - <<PY
import json
try:
    r=json.loads([l for l in open("gpurun_out/bench_gloo$N.json").read().splitlines() if l.startswith("{")][-1])
    print({k:r[k] for k in ("value","n_gpus","ms_per_step")}, r["sharded_vs_single_gpu_max_abs_delta"]["per_mode"], r["rank_census"]["ranks_seen"], r["exchange_timing"].get("form"), {k:v for k,v in r["cfg5"].items() if k in ("value","ms_per_step","error")}, r.get("graph_replay"), r["collectives_per_step"])
except Exception as e:
    print("no json:", e)
PY
  tail -3 gpurun_out/bench_gloo${N}.err | cut -c1-300
done
