# round 5, first GPU call: the sparse-coefficient-list path — parity tests, then the quick bench with and without it on the same box
ulimit -c 0
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse or flat_passgroup or corrupt_frame or config3 or batch_equals or subflights or concurrent_contexts or large_varblocks" 2>&1 | tail -15 > gpurun_out/r5a/pytest_sparse.txt; tail -5 gpurun_out/r5a/pytest_sparse.txt
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "single", c["single_frame_latency_ms"], "h2d", c.get("h2d_included_MPps"), "sparse ctx", c.get("contexts_on_sparse_coefficient_lists"), "dense retries", c.get("flights_repeated_with_dense_coefficients"))
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
for mode in sparse dense sparse2 dense2; do
  if [ "${mode#dense}" != "$mode" ]; then export JXLAMD_SPARSE=0; else unset JXLAMD_SPARSE; fi
  timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 2>gpurun_out/r5a/bench_${mode}_err.txt | tail -1 > gpurun_out/r5a/bench_$mode.json; echo $mode; show gpurun_out/r5a/bench_$mode.json
done
