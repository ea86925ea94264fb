# fresco_attn_f32 inside the flow network: tests + per-shape launch times of two forwards: bash tools/ab_a32_xcd.sh
# (round 6 used it with FRESCO_A32_XCD_MAP=1 / 0 -- an XCD-aware workgroup order that measured nothing and was removed:
#  profiles/r06_ab_attn32_forms.txt)
python -m pytest tests/test_gpu_attn32.py tests/test_gmflow.py tests/test_gpu_paras.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for X in 1 2; do python tools/bench_gmflow.py 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('forward', r['gmflow_forward_ms'], 'paras', r['get_flow_and_interframe_paras_ms'], 'attention', r['attention_ms_of_forward'])
for k,v in r['attn_f32_launches'].items(): print('   ', k, v['launches'], v['avg_us'], v['frac_executed_of_fp16_peak'])"; done
