# counters of the resonator kernel (k_bank_modes) on config 5, every mode live: separate --pmc passes, no trace options beside them
#   bash tools/bank_pmc.sh   (GPU box, from the repo root) -> gpurun_out/r04_bank_pmc.txt
ROOT=$PWD; mkdir -p $ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/bank_pmc_$i -- python3 $ROOT/tools/bank_bench.py --blocks 24 > /tmp/bank_pmc_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_kernel.py "k_bank_modes" /tmp/bank_pmc_* > gpurun_out/r04_bank_pmc.txt 2>&1
cat gpurun_out/r04_bank_pmc.txt
