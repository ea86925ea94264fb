#!/bin/bash
# A/B build + timing of kernel variants in ONE GPU-box visit (each variant = one object file rebuilt with extra flags,
# linked with the default objects into fresco_amd/lib/variants/libfresco_hip_<name>.so, selected via FRESCO_HIP_LIB).
# usage (repo root, on the box):  bash tools/ab_variants.sh [tag] ["stages"]   -> gpurun_out/ab_<tag>.txt
# (all stages: roughly 25 minutes of box time; stages: flash parity persist linear opt)
# Variants (name | source | flags):
#   base        the shipped build
#   noslp       attn.hip  -fno-slp-vectorize        no v_pk_mul_f32 in the exact-scale / rescale passes (packed f32 VALU is
#                                                   an anti-lever beside MFMAs: MI355X_MICROARCH.md)
#   foldinf     attn.hip  -DFOLD_MAX=1e9f           scale always folded: separates the exact pass from the data-dependent clock
#   fold0       attn.hip  -DFOLD_MAX=0.f            scale never folded: the exact pass on the bench's own activations
#   nomax15     attn.hip  -DNOMAX_THR=15.f
#   epiwide     attn.hip  -DFRESCO_EPI_WIDE=1       16-byte epilogue stores (v_permlane32_swap pairs)
#   priostat    attn.hip  -DFRESCO_PRIO_STATIC=1    waves 4-7 at s_setprio 1 for the whole loop, no per-segment flips
#   earlydma    attn.hip  -DFRESCO_EARLY_DMA=1      first key packs requested before the Q rows are loaded
#   combo       attn.hip  epiwide + priostat + earlydma
#   persist1/2  attn.hip  -DFRESCO_PERSIST=1 / 2    one workgroup per CU walks its query blocks; 2: the next block's first
#                                                   packs and Q rows are requested before the current epilogue
#   persist2e   attn.hip  persist2 + epiwide
#   pf4 / pf6   proj.hip  -DFRESCO_PROJ_PF=4 / 6    weight fragments read 4 / 6 MFMAs ahead
#   w4b2        proj.hip  -DFRESCO_PROJ_NWV=4 -DFRESCO_PROJ_NBUF=2   128-row workgroups, 2-slot ring (70 KB of LDS): TWO
#                                                   workgroups per CU whose x loads / epilogues overlap the other's MFMAs
#   w4b2pf4     both
#   adam2/adam4 opt.hip   -DFRESCO_ADAM_NOCT=2 / 4    adam_update: 2 / 4 channel octets per thread (the pixel's CSR rows are
#                                                   fetched once per thread, not once per octet)
#   gflush2     opt.hip   -DFRESCO_GRAM_FLUSH2=1      gram16w: sign-tile stores enumerated chunk-major (1 KB runs per wave store
#                                                   in the pre-tiled layout instead of 32-byte runs)
TAG=${1:-r}
OUT=$PWD/gpurun_out/ab_$TAG.txt
mkdir -p gpurun_out fresco_amd/lib/variants fresco_amd/csrc/build_var
make -C fresco_amd/csrc > /dev/null || exit 1
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
build() {  # name source per-file-flags variant-flags
  local name=$1 src=$2 extra=$3 var=$4
  local obj=fresco_amd/csrc/build_var/${src%.hip}_$name.o
  local so=fresco_amd/lib/variants/libfresco_hip_$name.so
  # (variant libraries built in the dev container travel with the snapshot: rebuild only when a source is newer)
  if [ -f $so ] && [ -z "$(find fresco_amd/csrc -maxdepth 1 \( -name '*.hip' -o -name '*.h' \) -newer $so)" ]; then return 0; fi
  $HIPCC $BASE $extra $var -c fresco_amd/csrc/$src -o $obj || return 1
  local objs=""
  for o in common attn attn32 proj temporal warp opt mapping; do
    if [ "$o.hip" = "$src" ]; then objs="$objs $obj"; else objs="$objs fresco_amd/csrc/build/$o.o"; fi
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o fresco_amd/lib/variants/libfresco_hip_$name.so
}
AT="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"
PR="-mllvm -amdgpu-mfma-vgpr-form"
build noslp attn.hip "$AT" "-fno-slp-vectorize"
build foldinf attn.hip "$AT" "-DFOLD_MAX=1e9f"
build fold0 attn.hip "$AT" "-DFOLD_MAX=0.f"
build nomax15 attn.hip "$AT" "-DNOMAX_THR=15.f"
build epiwide attn.hip "$AT" "-DFRESCO_EPI_WIDE=1"
build priostat attn.hip "$AT" "-DFRESCO_PRIO_STATIC=1"
build earlydma attn.hip "$AT" "-DFRESCO_EARLY_DMA=1"
build combo attn.hip "$AT" "-DFRESCO_EPI_WIDE=1 -DFRESCO_PRIO_STATIC=1 -DFRESCO_EARLY_DMA=1"
build persist1 attn.hip "$AT" "-DFRESCO_PERSIST=1"
build persist2 attn.hip "$AT" "-DFRESCO_PERSIST=2"
build persist2e attn.hip "$AT" "-DFRESCO_PERSIST=2 -DFRESCO_EPI_WIDE=1"
build pf4 proj.hip "$PR" "-DFRESCO_PROJ_PF=4"
build pf6 proj.hip "$PR" "-DFRESCO_PROJ_PF=6"
build w4b2 proj.hip "$PR" "-DFRESCO_PROJ_NWV=4 -DFRESCO_PROJ_NBUF=2"
build w4b2pf4 proj.hip "$PR" "-DFRESCO_PROJ_NWV=4 -DFRESCO_PROJ_NBUF=2 -DFRESCO_PROJ_PF=4"
build adam2 opt.hip "" "-DFRESCO_ADAM_NOCT=2"
build adam4 opt.hip "" "-DFRESCO_ADAM_NOCT=4"
build gflush2 opt.hip "" "-DFRESCO_GRAM_FLUSH2=1"
run() {  # name command...
  local name=$1; shift
  local lib=""
  [ "$name" != base ] && lib=$PWD/fresco_amd/lib/variants/libfresco_hip_$name.so
  echo "== $name: $*" >> $OUT
  FRESCO_HIP_LIB=$lib timeout 300 "$@" >> $OUT 2>&1
}
: > $OUT
STAGES=${2:-"flash parity persist linear opt"}   # second argument: a subset of the stages, e.g. "flash linear"
for st in $STAGES; do case $st in
flash)    # timing only (~15 s per line)
  for v in base noslp foldinf nomax15; do
    run $v python tools/bench_flash.py 20 1.0      # N(0,1) q, k: the cfg2c regime
  done
  for v in base noslp fold0 epiwide priostat earlydma combo persist1 persist2 persist2e; do
    run $v python tools/bench_flash.py 20 0.3      # small logits: the headline regime
  done ;;
parity)   # the attention variants against the test-suite (fold0 is exact by construction)
  FRESCO_TEST_QUEUED=1 run base python -m pytest tests/test_gpu_attention.py -q -p no:cacheprovider -k decoder_head_dims
  for v in noslp foldinf nomax15 epiwide earlydma combo; do
    run $v python -m pytest tests/test_gpu_attention.py -q -x -p no:cacheprovider
  done ;;
persist)  # several blocks per workgroup: full-size shapes + many-block fuzz
  for v in persist1 persist2 persist2e; do
    run $v python -m pytest tests/test_gpu_attention.py tests/test_gpu_fullsize.py -q -x -p no:cacheprovider
    run $v python tools/fuzz_attn.py 24 1 big
  done ;;
linear)
  for v in base pf4 pf6 w4b2 w4b2pf4; do
    run $v python tools/bench_linear.py
  done
  for v in pf4 pf6 w4b2 w4b2pf4; do
    run $v python -m pytest tests/test_gpu_linear.py -q -x -p no:cacheprovider
  done ;;
opt)
  for v in base adam2 adam4 gflush2; do
    run $v python tools/bench_opt.py 20 --no-baselines
  done
  for v in adam2 adam4 gflush2; do
    run $v python -m pytest tests/test_gpu_opt.py -q -x -p no:cacheprovider
  done ;;
esac; done
grep -E "^==|small-M HW=|spatial  HW=1024|q,k,v|passed|failed|all .* cases ok|FAIL|cfg3 extra" $OUT
