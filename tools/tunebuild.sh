# Rebuilds libunflow_hip.so with extra -D flags on the GPU box, runs tune.sh settings, one line per build.
# usage: tools/tunebuild.sh "<EXTRA flags>|<tune settings...>" ...
for spec in "$@"; do
  extra="${spec%%|*}"; rest="${spec#*|}"
  rm -f unflow_amd/csrc/correlation_mfma.o
  make -C unflow_amd/csrc EXTRA="$extra" > gpurun_out/build_tmp.log 2>&1 || { echo "BUILD FAILED: $extra"; tail -5 gpurun_out/build_tmp.log; continue; }
  echo "== EXTRA='$extra'"
  bash tools/tune.sh $rest
done
