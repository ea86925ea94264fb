# A/B of fresco_attn_f32's workgroup order (a problem's query blocks on ONE XCD vs spread over all eight): bash tools/ab_a32_xcd.sh
python -m pytest tests/test_gpu_attn32.py tests/test_gmflow.py tests/test_gpu_paras.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for X in 1 0 1 0; do FRESCO_A32_XCD_MAP=$X python tools/bench_gmflow.py 2>&1 | tail -1 | X=$X python -c "
import json,sys,os
r=json.loads(sys.stdin.read())
print('a32 xcd map', os.environ['X'], 'forward', r['gmflow_forward_ms'], 'paras', r['get_flow_and_interframe_paras_ms'], 'attention', r['attention_ms_of_forward'])
for k,v in r['attn_f32_launches'].items(): print('   ', k, v['launches'], v['avg_us'], v['frac_executed_of_fp16_peak'])"; done
