# round 5: BASELINE configs[3] on ONE MI355X — a 32768 x 32768 frame as 8 concurrent bands (bench line + band phase trace), and the C-ABI form of the same decode
ulimit -c 0
mkdir -p gpurun_out/r5l
JXLAMD_TRACE_BANDS=1 timeout 2700 python bench.py --workload c4 --steps 3 --warmup 1 2> gpurun_out/r5l/c4_trace.txt | tail -1 > gpurun_out/r5l/bench_c4.json
cut -c1-400 gpurun_out/r5l/bench_c4.json; grep "bands rank" gpurun_out/r5l/c4_trace.txt | tail -4
