# A/B of the small-plane Gram forms (round 6): usage (repo root on the GPU box): bash tools/ab_small.sh
export BENCH_OPT_LAYERS=0,1
run() { echo "== $1"; shift; env "$@" python tools/bench_opt.py 20 --no-baselines 2>&1 | grep -E "^layer" | cut -c1-130; }
run "round-5 forms (lane-per-row fragment loads; one launch at 8x8)" FRESCO_GRAM_SPLIT_WG=0 FRESCO_GRAM_COOP=0
run "8x8: split-K slices on their own CUs + ordered sum; 16x16: round 5" FRESCO_GRAM_SPLIT_WG=1 FRESCO_GRAM_COOP=0
run "cooperative coalesced staging (gram16c), 8x8 one launch" FRESCO_GRAM_SPLIT_WG=0 FRESCO_GRAM_COOP=1
run "cooperative coalesced staging at 16x16, split workgroups at 8x8 (default)" FRESCO_GRAM_SPLIT_WG=1 FRESCO_GRAM_COOP=1
python -m pytest tests/test_gpu_opt.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "cooperative or split_workgroup or launch_forms or closure" 2>&1 | tail -3
