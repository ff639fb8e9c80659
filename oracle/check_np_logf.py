"""All 2^32 float32 bit patterns: oracle/np_logf.c (the restatement of numpy's SIMD float32 log) against np.log of the
installed numpy.  Test infrastructure; run by hand:  python oracle/check_np_logf.py  (about a minute on 8 cores).
Prints one JSON line; profiles/r04_np_logf_exhaustive.txt keeps the build container's."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def main():
    t0 = time.time()
    bad, first = 0, None
    piece = 1 << 26
    with np.errstate(all="ignore"):
        for start in range(0, 1 << 32, piece):
            x = np.arange(start, start + piece, dtype=np.uint32).view(np.float32)
            diff = O.c_np_logf(x).view(np.uint32) != np.log(x).view(np.uint32)
            n = int(np.count_nonzero(diff))
            if n and first is None:
                first = hex(start + int(np.argmax(diff)))
            bad += n
    from numpy._core._multiarray_umath import __cpu_features__ as feats

    print(json.dumps({"patterns": 1 << 32, "mismatches": bad, "first_mismatch": first, "numpy": np.__version__,
                      "cpu_features": [k for k in ("FMA3", "AVX2", "AVX512F", "AVX512_SKX") if feats.get(k)], "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
