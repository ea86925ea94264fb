for i in 1 2; do for F in 0 1; do echo "== FUSE_QKV=$F"; FRESCO_GMFLOW_FUSE_QKV=$F python tools/bench_gmflow.py 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
d=r['dense_layers']
print(r['gmflow_forward_ms'], r['get_flow_and_interframe_paras_ms'], d['ms_of_forward'], {k:(v['launches'],v['avg_us']) for k,v in d['per_shape'].items() if 'K128_kh0' in k})"; done; done
