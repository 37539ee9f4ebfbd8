# round 5, last call: the new regression test, the mixed line at the step count of the round's earlier record (8 / 2) and at the defaults' (20 / 5), the driver's bench command on the final code
ulimit -c 0
O=gpurun_out/aj; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu -k "misses_the_lf_table_pool or composed_frames_ride or concurrent_contexts" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --workload mixed --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_mixed_s8_$i.json; cut -c1-200 $O/bench_mixed_s8_$i.json; done
timeout 600 python bench.py --workload mixed --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_mixed_s20.json; cut -c1-200 $O/bench_mixed_s20.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_err.txt | tail -1 > $O/bench_driver_cmd.json; cut -c1-200 $O/bench_driver_cmd.json
