# Rebuilds one source with extra -D flags on the GPU box and prints a tune.sh line per build (A/B of compile-time knobs).
# usage: tools/tunebuild.sh <file.hip> "<EXTRA flags>" ...
src=$1; shift
for extra in "$@"; do
  rm -f unflow_amd/csrc/${src%.hip}.o
  make -C unflow_amd/csrc EXTRA="$extra" > gpurun_out/build_tmp.log 2>&1 || { echo "BUILD FAILED: $extra"; tail -5 gpurun_out/build_tmp.log; continue; }
  echo "== EXTRA='$extra'"
  bash tools/tune.sh base
done
