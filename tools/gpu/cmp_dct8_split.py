import sys, numpy as np, os, subprocess
sys.path.insert(0, "tests")
from conftest import load_case, VARDCT_CASES
import jxl_coder_amd as J
if len(sys.argv) > 1:
    dec = J.JxlDecoder(0); ref = np.load("/tmp/split1.npz")
    for n in VARDCT_CASES:
        o = dec.decode_one_shot(load_case(n)[0])[0]; d = np.abs(o.astype(int) - ref[n].astype(int))
        print(n, "differing samples", int((d > 0).sum()), "of", d.size, "max", int(d.max()))
else:
    dec = J.JxlDecoder(0)
    np.savez("/tmp/split1.npz", **{n: dec.decode_one_shot(load_case(n)[0])[0] for n in VARDCT_CASES})
    print(subprocess.run([sys.executable, "tools/gpu/cmp_dct8_split.py", "x"], env=dict(os.environ, JXLAMD_DCT8_SPLIT="0"), capture_output=True, text=True).stdout)
