#!/bin/bash
# Instruction-cache counters of the LF kernel: alone (single 4K frame: 4 LF waves on the chip) and in a flight of 128 frames (512 LF waves, 2 per CU)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INST_CYCLES_VMEM\|SQ_WAIT_ANY" | sort -u) > $R/gpurun_out/icache_avail.txt
cat $R/gpurun_out/icache_avail.txt
export PMC_SETS="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES;SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
bash $R/tools/gpu/run_pmc.sh > $R/gpurun_out/icache_single.txt 2>&1
bash $R/tools/gpu/run_pmc_batch.sh > $R/gpurun_out/icache_flight128.txt 2>&1
grep -i "lf_group" $R/gpurun_out/icache_single.txt $R/gpurun_out/icache_flight128.txt | cut -c1-400
