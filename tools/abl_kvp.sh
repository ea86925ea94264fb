for v in 0 1 2; do FRESCO_HIP_LIB=$PWD/fresco_amd/lib/abl_kvp$v.so python tools/bench_kvproj.py 2>&1 | grep -E "lib|fused" ; done
python -m pytest tests/test_gpu_attention.py -m gpu -q -k fused -p no:cacheprovider 2>&1 | tail -2
