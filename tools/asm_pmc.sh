# counters of the assembly kernel (separate --pmc passes, no trace options beside them)
ROOT=$PWD; mkdir -p $ROOT/gpurun_out/r02f; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/asm_pmc_$i -- python3 $ROOT/tools/assembly_bench.py > $ROOT/gpurun_out/r02f/pmc_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_kernel.py "k_assemble_flat<10" /tmp/asm_pmc_* > gpurun_out/r02f/asm_pmc.txt 2>&1
cat gpurun_out/r02f/asm_pmc.txt
