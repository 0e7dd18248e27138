"""Engine-op rates on this GPU (the figures bench.py reports as `extra`), without the NTT / CPU legs:
    python tools/eo.py [quick] [--ext-cols-max K]"""
import json
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import bench  # noqa: E402
import __graft_entry__ as g  # noqa: E402

g.build()
if "--ext-cols-max" in sys.argv:   # largest logN - 12 whose extension runs as the column kernel (lf_tune)
    from liberate_fhe_amd._native import lib
    lib.lf_tune(1, int(sys.argv[sys.argv.index("--ext-cols-max") + 1]))
rates, roof = bench.engine_rates("cuda:0", quick=len(sys.argv) > 1 and sys.argv[1] == "quick")
print(json.dumps({k: round(v, 1) for k, v in rates.items()}))
print(json.dumps({k: round(v["frac"], 4) for k, v in roof.items()}))
