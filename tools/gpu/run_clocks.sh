# Sample the shader clock / power / busy % while bench.py runs.  Output: gpurun_out/clocks.log
ulimit -c 0
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 24 --warmup 2 > gpurun_out/clocks_bench.log 2>&1 &
BP=$!
sleep 25
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use|mclk|fclk" | tr '\n' ';' | sed 's/  */ /g'; echo
  sleep 1
done > gpurun_out/clocks.log 2>&1
wait $BP
tail -1 gpurun_out/clocks_bench.log | cut -c1-120
cat gpurun_out/clocks.log | cut -c1-400
echo "--- idle"; sleep 3
rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | tr '\n' ';' | sed 's/  */ /g'
