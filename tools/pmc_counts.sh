# usage: bash tools/pmc_counts.sh <name> [<so>] : instruction counts of the correlation backward of one build (one PMC pass)
name=$1; so=$2
R=$GRAFT_REPO_ROOT
export UNFLOW_ALLOW_DIAGNOSTIC_BUILD=1
if [ -n "$so" ]; then export UNFLOW_HIP_LIB=$R/$so; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmcc_${name} -- python $R/tools/corr_only.py 32 bwd 2 > $R/gpurun_out/pmcc_${name}.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmcc_${name}/ > gpurun_out/pmcc_${name}.txt; echo "== $name"; grep "SQ_" gpurun_out/pmcc_${name}.txt
