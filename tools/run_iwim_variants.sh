export UNFLOW_ALLOW_DIAGNOSTIC_BUILD=1   # the variants are diagnostic builds (unflow_build_flags())
mkdir -p gpurun_out
out=gpurun_out/iwim_variants_${1:-run}.txt
timeout -k 10 200 python tools/iwim_variant.py default >> $out 2>gpurun_out/variants_err.log || echo "default FAILED" >> $out
for so in unflow_amd/variants/*.so; do
  n=$(basename $so .so)
  timeout -k 10 200 python tools/iwim_variant.py $n $so >> $out 2>>gpurun_out/variants_err.log || echo "$n FAILED" >> $out
done
cat $out
