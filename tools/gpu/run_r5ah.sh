# round 5, last measurements on the final code: pixel hashes of three RGBA frames, the -m gpu suite, the mixed-content evidence (run_profiles_mixed.sh), the driver's
# bench command, the two Modular timing tools
ulimit -c 0
O=gpurun_out/ah; mkdir -p $O
sed -i 's/for v in "" _pack _pack2; do/for v in ""; do/' tools/gpu/run_r5ae.sh
bash tools/gpu/run_r5ae.sh 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
( time timeout 1300 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
bash tools/gpu/run_profiles_mixed.sh > $O/mixed.txt 2>&1; tail -30 $O/mixed.txt | cut -c1-220
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_err.txt | tail -1 > $O/bench_driver_cmd.json; cut -c1-260 $O/bench_driver_cmd.json
timeout 600 python tools/gpu/prev_channel_time.py > $O/prev_channel.txt 2>&1; tail -4 $O/prev_channel.txt | cut -c1-220
timeout 600 python tools/gpu/lossy_palette_time.py > $O/lossy_palette.txt 2>&1; tail -4 $O/lossy_palette.txt | cut -c1-220
