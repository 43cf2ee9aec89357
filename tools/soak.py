#!/usr/bin/env python3
"""A longer run of tests/test_soak.py's randomised cases on the GPU (not part of the suite):  tools/soak.py [batches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from digiham_amd import api
from oracle import oracle as O
import test_soak as T
from common import assert_matches_oracle, run_engine
ctx = api.Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
done = 0
for proto in ("dmr", "ysf", "nxdn"):
    for batch in range(2, 2 + n):
        rng = np.random.default_rng(777000 + 31 * batch + {"dmr": 0, "ysf": 100000, "nxdn": 200000}[proto])
        for c in range(4):
            x, kw, okw, chunks, what = T._case(rng, proto)
            ref = O.chain(x[None, :], **okw)
            res = run_engine(ctx, x[None, :], proto, chunks, **kw)
            assert_matches_oracle(res, ref, 1, "%s batch %d case %d %r" % (proto, batch, c, what))
            done += 1
    print(proto, "ok", done, flush=True)
print("soak: %d cases bit-exact" % done)
