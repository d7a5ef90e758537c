# counters of the fp64 SpMM at w = 64 (tools/spmm_bench.py, 20 launches), separate --pmc passes, no trace options beside them
ROOT=$PWD; mkdir -p $ROOT/gpurun_out/r02h; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  WIDTHS=64 rocprofv3 --pmc $set --output-format csv -d /tmp/spmm_pmc_$i -- python3 $ROOT/tools/spmm_bench.py > $ROOT/gpurun_out/r02h/pmc_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_kernel.py "k_spmm_wide<double, double, double" /tmp/spmm_pmc_* > gpurun_out/r02h/spmm_pmc.txt 2>&1
cat gpurun_out/r02h/spmm_pmc.txt
