#!/bin/bash
# Regenerates the committed evidence under profiles/ on a 1-GPU MI355X box (run from the repo root, e.g. through gpurun):
#   default bench line, rocprofv3 kernel-trace summary of the same command, PMC HBM-traffic passes, MFMA utilisation of the prefill.
# PMC passes are separate runs with --kernel-trace only (no other trace domains), one counter group per pass.
set -e
# (the 'W through LDS' rows of the dl_linear_tiles tables need a library built with HIPCC_EXTRA=-DDL_LT_MEASURE python -m dynamic_llava_amd.build_ext --force; without it they are skipped)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
R=${1:-r01}
RAW=/tmp/dl_prof_raw   # raw rocprofv3 databases stay off gpurun_out/ (64 MiB merge limit): only the summaries go there
rm -rf "$RAW"; mkdir -p gpurun_out "$RAW"
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rocprofv3 --kernel-trace --stats -d $RAW/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-gpu --no-extra-legs > gpurun_out/bench_prof.json 2>/dev/null
python tools/prof_summary.py "$(find $RAW/prof_bench -name '*.db' | head -1)" 45 > gpurun_out/kernel_stats.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW/pmc_fetch -o p -- python tools/pmc_probe.py > gpurun_out/pmc_fetch.log 2>/dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW/pmc_write -o p -- python tools/pmc_probe.py > gpurun_out/pmc_write.log 2>/dev/null
python tools/pmc_report.py "$(find $RAW/pmc_fetch -name '*.db' | head -1)" "$(find $RAW/pmc_write -name '*.db' | head -1)" gpurun_out/pmc_fetch.log > gpurun_out/pmc_report.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $RAW/pmc_mfma -o m -- python tools/prefill_kernels.py > /dev/null 2>&1
python tools/mfma_report.py "$(find $RAW/pmc_mfma -name '*.db' | head -1)" > gpurun_out/mfma_report.txt
# round 4: the variable-shape request stream (VQAL:123-196) next to the same stream with one width; BASELINE configs[2] / [4]; the N > 1 path as two
# ranks on this one GPU (gloo; bench.py starts its own ranks)
python tools/bench_varlen_stream.py > gpurun_out/stream_var.json 2> gpurun_out/stream_var.err
python tools/bench_varlen_stream.py --fixed > gpurun_out/stream_fixed.json 2>/dev/null
python tools/bench_configs.py c2 c4 2>/dev/null | tail -1 > gpurun_out/bench_configs_2_4.json
python tools/bench_configs.py c4cal 2>/dev/null | tail -1 > gpurun_out/bench_configs_4_calibrated.json
DL_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_dp2_one_gpu.json 2> gpurun_out/bench_dp2_one_gpu.err
python bench.py --steps 10 --warmup 3 --new-tokens 128 --no-cpu-baseline --no-ref-gpu > gpurun_out/bench_b1_128tokens.json 2>/dev/null
# round 5: dl_linear_packed -- microbench table, per-wave timelines, request-counter passes per variant (library / splitk / packed variants), L2-hit ceilings
python tools/bench_linear_packed.py --m 170,117 --sweep > gpurun_out/linear_packed_bench.txt 2>/dev/null
for a in "qkv --nu 3 --ks 1" "qkv --nu 6 --ks 2" "gate|up --nu 6 --ks 1" "o --nu 2 --ks 2" "down --nu 4 --ks 4"; do python tools/lp_timeline.py --shape $a; done > gpurun_out/linear_packed_timelines.txt 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS -d $RAW/lp_sq -o s -- python tools/pmc_linear_packed_probe.py > gpurun_out/lp_probe.log 2>/dev/null
rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES TCC_BUSY TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_LEVEL TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY -d $RAW/lp_tc -o s -- python tools/pmc_linear_packed_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $RAW/lp_l2 -o s -- python tools/pmc_linear_packed_probe.py > /dev/null 2>&1
python tools/pmc_linear_packed_report.py gpurun_out/lp_probe.log $(find $RAW/lp_sq $RAW/lp_tc $RAW/lp_l2 -name '*.db') > gpurun_out/linear_packed_counters.txt 2>&1
(hipcc --offload-arch=gfx950 -O3 tools/l2_read_bw.hip -o /tmp/l2bw 2>/dev/null && /tmp/l2bw > gpurun_out/l2_read_ceilings.txt 2>&1) || true
# round 6: dl_linear_tiles (CLIP tower / projector GEMMs) -- microbench vs hipBLASLt with cold weights, per-wave timelines incl. the two-pass (L2-hit) and
# 32-workgroup experiments; the prefill's kernel inventory; MFMA-busy of the configs[2] prefill (the MFMA-bound leg)
python tools/bench_linear_tiles.py > gpurun_out/linear_tiles_bench.txt 2>/dev/null
python tools/bench_linear_tiles.py --stamps > gpurun_out/linear_tiles_timelines.txt 2>/dev/null
rocprofv3 --kernel-trace -d $RAW/prof_pf -o pf -- python tools/prefill_kernels.py > /dev/null 2>&1
python tools/prof_summary.py "$(find $RAW/prof_pf -name '*.db' | head -1)" 60 --after spin_kernel > gpurun_out/prefill_kernel_inventory.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $RAW/pmc_mfma_c2 -o m -- python tools/prefill_kernels_c2.py > /dev/null 2>&1
python tools/mfma_report.py "$(find $RAW/pmc_mfma_c2 -name '*.db' | head -1)" > gpurun_out/mfma_report_c2.txt
# then, back in the development container (gpurun merges gpurun_out/):
#   cp gpurun_out/bench_default.json profiles/${R}_bench_b1.json; cp gpurun_out/kernel_stats.txt profiles/${R}_bench_kernel_stats.txt
#   grep -v '^JSON' gpurun_out/pmc_report.txt > profiles/${R}_pmc_traffic.txt; grep '^JSON' gpurun_out/pmc_report.txt | sed 's/^JSON //' > profiles/${R}_pmc_traffic.json
#   grep -v '^JSON' gpurun_out/mfma_report.txt > profiles/${R}_prefill_mfma_util.txt
#   cp gpurun_out/stream_var.json profiles/${R}_varlen_stream.json; cp gpurun_out/stream_fixed.json profiles/${R}_varlen_stream_fixed_width.json
#   cp gpurun_out/linear_packed_{bench,timelines,counters}.txt gpurun_out/l2_read_ceilings.txt -> profiles/${R}_*
#   cp gpurun_out/bench_configs_2_4.json profiles/${R}_bench_configs_2_4.json; cp gpurun_out/bench_dp2_one_gpu.json profiles/${R}_bench_dp2_one_gpu.json
#   cp gpurun_out/linear_tiles_{bench,timelines}.txt gpurun_out/prefill_kernel_inventory.txt -> profiles/${R}_*; grep -v '^JSON' gpurun_out/mfma_report_c2.txt > profiles/${R}_configs2_prefill_mfma_util.txt
echo "done: gpurun_out/{bench_default.json,kernel_stats.txt,pmc_report.txt,mfma_report.txt}"
