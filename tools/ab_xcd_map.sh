# A/B of fn_gemm's workgroup order (XCD-aware column-block walk vs column block outermost): bash tools/ab_xcd_map.sh
python -m pytest tests/test_gpu_flownet.py tests/test_gmflow.py tests/test_gpu_paras.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for X in 1 0; do echo "FRESCO_FN_XCD_MAP=$X"; FRESCO_FN_XCD_MAP=$X python tools/ubench_fn_gemm.py 2>&1 | grep -E "^M=65536 K=(128 N=384|256 N=1024|128 N=128|1024 N=128)" | cut -c1-160; done
for X in 1 0 1 0; do FRESCO_FN_XCD_MAP=$X python tools/bench_gmflow.py 2>&1 | tail -1 | X=$X python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); d=r['dense_layers']
print('xcd map', os.environ['X'], 'forward', r['gmflow_forward_ms'], 'paras', r['get_flow_and_interframe_paras_ms'], 'dense', d['ms_of_forward'])
for k,v in d['per_shape'].items():
    if any(t in k for t in ('N1024','N384','N256','N576')): print('   ', k, v)"; done
