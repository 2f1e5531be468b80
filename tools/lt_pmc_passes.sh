#!/bin/bash
# Counter passes over tools/pmc_linear_tiles_probe.py (one counter group per pass, --kernel-trace only): what bounds dl_linear_tiles' k loop
set -e
# (the 'W through LDS' rows of the dl_linear_tiles tables need a library built with HIPCC_EXTRA=-DDL_LT_MEASURE python -m dynamic_llava_amd.build_ext --force; without it they are skipped)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
RAW=/tmp/dl_lt_raw
rm -rf "$RAW"; mkdir -p gpurun_out "$RAW"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS -d $RAW/sq -o s -- python tools/pmc_linear_tiles_probe.py > gpurun_out/lt_probe.log 2>/dev/null
rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES TCC_BUSY TCC_EA0_RDREQ TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TA_BUSY -d $RAW/tc -o s -- python tools/pmc_linear_tiles_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $RAW/l2 -o s -- python tools/pmc_linear_tiles_probe.py > /dev/null 2>&1
python tools/pmc_linear_packed_report.py gpurun_out/lt_probe.log $(find $RAW/sq $RAW/tc $RAW/l2 -name '*.db') > gpurun_out/linear_tiles_counters.txt 2>&1
