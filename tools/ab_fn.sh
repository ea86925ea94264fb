python tools/ubench_fn_gemm.py 2>&1 | grep -E "^M=65536|^M=16384 K=128 N=128" | cut -c1-150
python -m pytest tests/test_gpu_flownet.py tests/test_gmflow.py tests/test_gpu_paras.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do python tools/bench_gmflow.py 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); d=r['dense_layers']
print(r['gmflow_forward_ms'], r['get_flow_and_interframe_paras_ms'], d['ms_of_forward'], r['attention_ms_of_forward'], r['gmflow_forward_library_ops_ms'])"; done
python -m pytest tests/test_gpu_opt.py tests/test_gpu_cfg45.py -m gpu -q -p no:cacheprovider -k "not full_batch and not 32_frames" 2>&1 | tail -2
BENCH_OPT_LAYERS=0,1 python tools/bench_opt.py 20 --no-baselines 2>&1 | grep -E "^layer" | cut -c1-130
