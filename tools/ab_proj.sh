for L in abl_projdirect.so libfresco_hip.so; do echo "== $L"; FRESCO_HIP_LIB=$PWD/fresco_amd/lib/$L python bench.py --no-cpu-baseline --no-aux 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_all'])
print({k:v for k,v in r['kernel_avg_us'].items() if k.startswith('linear')})"; done
python -m pytest tests/test_gpu_linear.py tests/test_gpu_attention.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
