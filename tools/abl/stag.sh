for v in base stag1 stag2 stag3; do
  lib=""; [ $v != base ] && lib=$PWD/tools/abl/libfresco_hip_$v.so
  echo "== $v"; FRESCO_HIP_LIB=$lib timeout 120 python tools/bench_linear.py 2>&1 | head -2
done
timeout 300 python -m pytest tests/test_gpu_attention.py -q -x -p no:cacheprovider -k "underfilled or plain or logit" 2>&1 | tail -2
timeout 200 python tools/fuzz_attn.py 40 3 2>&1 | tail -1
