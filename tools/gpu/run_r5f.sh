# round 5: DCT128 / DCT256 families on the GPU (hand-written codestreams vs the reference's pixels), singles and inside flights
ulimit -c 0
mkdir -p gpurun_out/r5f
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "w_dct or flat_passgroup or sparse or batch_of_round" 2>&1 | tail -8
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import load_case, WRITER_CASES
import jxl_coder_amd as J
dec = J.JxlDecoder(0)
names = WRITER_CASES + ["v264x520_e7", "asset_first_jxl"]
datas = [load_case(n)[0] for n in names]
singles = [dec.decode_one_shot(d)[0] for d in datas]
for n, s in zip(names, singles):
    e = load_case(n)[1]; d = np.abs(s.astype(int) - e.astype(int)); print(n, "max", d.max(), "mean", round(float(d.mean()), 5))
for rep in range(3):
    outs = [torch.zeros(s.size, dtype=torch.uint8, device="cuda") for s in singles]
    dec.decode_batch_to_device(datas, [o.data_ptr() for o in outs], [o.numel() for o in outs])
    torch.cuda.synchronize()
    print("flight", rep, [bool(np.array_equal(o.cpu().numpy().reshape(s.shape), s)) for s, o in zip(singles, outs)])
PY
