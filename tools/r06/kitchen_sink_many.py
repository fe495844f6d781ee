"""r06: tests/test_gpu_kitchen_sink.py's mixed worlds for many more seeds than the suite runs (python tools/r06/kitchen_sink_many.py <first> <last>)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from tests.test_gpu_kitchen_sink import test_mixed_worlds_against_the_oracle as run
ctx = mgf_amd.Context(0)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    try:
        run(ctx, seed)
    except AssertionError as e:
        msg = str(e).splitlines()[0][:200] if str(e) else "assert"
        if msg.strip().isdigit() or "assert" in msg and ">" in msg: print("seed", seed, "sparse scene (peak", msg, ")"); continue
        bad.append(seed); print("seed", seed, "FAILED:", msg); traceback.print_exc(limit=2)
    except mgf_amd.MgfError as e:
        print("seed", seed, "library error:", str(e)[:160])
print("seeds", a, "..", b, ":", len(bad), "parity failures", bad)
ctx.close()
