import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0,'/root/repo')
import tests.test_gpu_round4 as T
from datasketch_amd import WeightedMinHashGenerator, _native
dim, s = int(sys.argv[1]), int(sys.argv[2])
import zlib
rng = np.random.RandomState(zlib.crc32(f"{dim}/{s}".encode()))
n = 3 * 2048 + 37
x = T._fuzz_matrix(rng, n, dim, (dim + s) % 2 == 1)
ctx = _native.context()
for mode in (False, True):
    g = WeightedMinHashGenerator(dim, s, seed=11, gpu_mode="always", device_log=mode)
    for kern in (0, 1):
        ctx.set_option("weighted.kernel", kern)
        out, ne = g.minhash_many_arrays(x)
        ctx.set_option("weighted.kernel", 0)
        csr = sp.csr_matrix(x)
        want, wn = T._oracle_rows(g, csr, np.arange(n))
        bad = [i for i in range(n) if wn[i] and i not in (5, 6, n-3) and not np.array_equal(out[i], want[i])]
        print("device_log", mode, "kernel", kern, "bad rows", len(bad), bad[:10], "ne ok", np.array_equal(ne.astype(bool), wn))
        for i in bad[:3]:
            nz = np.count_nonzero(x[i]); d = np.argwhere(out[i] != want[i])
            print("  row", i, "wave", i % 2048, "turn", i // 2048, "prev row stored", np.count_nonzero(x[i - 2048]) if i >= 2048 else None, "stored", nz, "max", x[i].max(), "diff samples", len(d), d[:4].tolist(), out[i][d[0][0]], want[i][d[0][0]])
# log check
v = x[x > 0][:200000]
print("log equal", np.array_equal(ctx.weighted_logf(v).view(np.uint32), np.log(v).view(np.uint32)))
