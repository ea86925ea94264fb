# A/B of the 3 x 3 convolution forms of csrc/flownet.hip (window in LDS vs im2col view): bash tools/ab_conv_patch.sh
python -m pytest tests/test_gpu_flownet.py tests/test_gmflow.py tests/test_gpu_paras.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for P in 1 0 1 0; do FRESCO_FN_CONV_PATCH=$P python tools/bench_gmflow.py 2>&1 | tail -1 | P=$P python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); d=r['dense_layers']
print('patch', os.environ['P'], 'forward', r['gmflow_forward_ms'], 'paras', r['get_flow_and_interframe_paras_ms'], 'dense', d['ms_of_forward'])
for k,v in d['per_shape'].items():
    if 'kh3' in k: print('   ', k, v)"; done
