python -m pytest tests/test_gpu_flownet.py tests/test_gmflow.py tests/test_gpu_paras.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do python tools/bench_gmflow.py 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); d=r['dense_layers']
print('forward', r['gmflow_forward_ms'], 'paras', r['get_flow_and_interframe_paras_ms'], 'dense', d['ms_of_forward'], 'attention', r['attention_ms_of_forward'])
for k,v in d['per_shape'].items(): print('   ', k, v)"; done
