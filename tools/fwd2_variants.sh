# Builds variants of the second-generation correlation forward HERE (they travel with gpurun):
#   tools/fwd2_variants.sh name1 "FWD2_KNOB=1 ..." "-DCXXFLAG ..."  name2 "..." "..."
# generator knobs (environment of tools/gen_corr_fwd_asm.py) + compiler flags -> unflow_amd/variants/<name>.so
set -e
cd "$(dirname "$0")/../unflow_amd/csrc"
mkdir -p ../variants
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed -Wno-inline-asm"
make -s >/dev/null
while [ $# -gt 2 ]; do
  name=$1; knobs=$2; extra=$3; shift 3
  inc=$(pwd)/../variants/$name.inc
  env $knobs FWD2_OUT=$inc python ../../tools/gen_corr_fwd_asm.py >/dev/null
  $HIPCC $FLAGS $extra -DUNFLOW_FWD2_INC="\"$inc\"" -c -o ../variants/$name.o correlation_fwd2.hip
  objs=""
  for o in api warp census stencil stacked correlation correlation_mfma correlation_fwd2 correlation_bwd4 correlation_f16; do
    if [ "$o" = "correlation_fwd2" ]; then objs="$objs ../variants/$name.o"; else objs="$objs $o.o"; fi
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so $objs
  rm -f ../variants/$name.o ../variants/$name.inc
  echo "built variants/$name.so  ($knobs $extra)"
done
