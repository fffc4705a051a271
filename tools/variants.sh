# Builds A/B variants of one kernel source HERE (hipcc cross-compiles; the .so files travel with gpurun):
#   tools/variants.sh correlation_mfma.hip name1 "-DKNOB=1" name2 "-DKNOB=2 -DOTHER" ...
# -> unflow_amd/variants/<name>.so ; run them on the GPU box with tools/run_variants.sh
set -e
src=$1; shift
cd "$(dirname "$0")/../unflow_amd/csrc"
mkdir -p ../variants
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed"
make -s >/dev/null
while [ $# -gt 1 ]; do
  name=$1; extra=$2; shift 2
  $HIPCC $FLAGS $extra -c -o ../variants/$name.o $src
  objs=""
  for o in api warp census stencil stacked correlation correlation_mfma correlation_fwd2 correlation_bwd4 correlation_f16; do
    if [ "$o.hip" = "$src" ]; then objs="$objs ../variants/$name.o"; else objs="$objs $o.o"; fi
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so $objs
  rm -f ../variants/$name.o
  echo "built variants/$name.so  ($extra)"
done
