# counters of the tall-skinny fp64 MFMA kernels (tools/dense_bench.py), separate --pmc passes, no trace options beside them
# usage (GPU box): bash tools/dense_pmc.sh <out-dir under gpurun_out>
ROOT=$PWD; OUT=$ROOT/gpurun_out/${1:-dense_pmc}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  rm -rf /tmp/dense_pmc_$i
  CASES=${CASES:-160+80,80+80} rocprofv3 --pmc $set --output-format csv -d /tmp/dense_pmc_$i -- python3 $ROOT/tools/dense_bench.py > $OUT/pmc_$i.log 2>&1
done
cd $ROOT
for k in "k_combine<10" "k_combine<6" "k_gram_blocked<5, 3, 1, 2, 2, 2>" "k_gram_blocked<5, 3, 1, 1, 4, 2>"; do echo "== $k"; python3 tools/pmc_kernel.py "$k" /tmp/dense_pmc_*; done > $OUT/dense_pmc.txt 2>&1
cat $OUT/dense_pmc.txt
