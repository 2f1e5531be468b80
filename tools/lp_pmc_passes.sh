set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
RAW=/tmp/dl_prof_raw; rm -rf $RAW; mkdir -p $RAW gpurun_out
python tools/bench_linear_packed.py --m 170,117,32 --sweep > gpurun_out/linear_packed_bench.txt 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS -d $RAW/lp_sq -o s -- python tools/pmc_linear_packed_probe.py > gpurun_out/lp_probe.log 2>/dev/null
rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES TCC_BUSY TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_LEVEL TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY -d $RAW/lp_tc -o s -- python tools/pmc_linear_packed_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $RAW/lp_l2 -o s -- python tools/pmc_linear_packed_probe.py > /dev/null 2>&1 || echo "l2 pass failed"
python tools/pmc_linear_packed_report.py gpurun_out/lp_probe.log $(find $RAW/lp_sq $RAW/lp_tc $RAW/lp_l2 -name '*.db') > gpurun_out/linear_packed_counters.txt 2>&1 || true
(hipcc --offload-arch=gfx950 -O3 tools/l2_read_bw.hip -o /tmp/l2bw 2>/dev/null && /tmp/l2bw > gpurun_out/l2_read_ceilings.txt 2>&1) || true
ls -la $RAW/*
