"""Extended schedule fuzz (not part of the suite): the bit-identity fuzz tests of tests/test_gpu_parity.py over many more seeds.
usage: fuzz_more.py first last"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from tests import test_gpu_parity as T
dev = torch.device("cuda:0")
a, b = int(sys.argv[1]), int(sys.argv[2])
ok = skip = 0
t0 = time.time()
for seed in range(a, b):
    for fn in (T.test_persistent_schedule_fuzz_bit_identical, T.test_same_group_launch_fuzz_bit_identical):
        try:
            fn(dev, seed)
            ok += 1
        except pytest.skip.Exception:
            skip += 1
        except Exception as e:
            print("FAILED", fn.__name__, seed, repr(e)[:300], flush=True)
            raise
print(f"seeds {a}..{b - 1}: {ok} passed, {skip} skipped in {time.time() - t0:.0f} s", flush=True)
