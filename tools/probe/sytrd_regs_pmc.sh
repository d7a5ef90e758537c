# counters of k_sytrd_regs at order 222, separate --pmc passes, no trace options beside them.   bash tools/probe/sytrd_regs_pmc.sh
ROOT=$PWD; OUT=$ROOT/gpurun_out/sytrd_regs_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/sr_pmc_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/sr_pmc_$i -- python3 $ROOT/tools/probe/sytrd_regs_order222.py > $OUT/pmc_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_kernel.py "k_sytrd_regs" /tmp/sr_pmc_* > $OUT/sytrd_regs_pmc.txt 2>&1
cat $OUT/sytrd_regs_pmc.txt
