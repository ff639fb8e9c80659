import sys, numpy as np
sys.path.insert(0, '/root/repo')
from datasketch_amd import WeightedMinHashGenerator, _native
n, dim, s = 3000, 4096, 128
rs = np.random.RandomState(42)
x = rs.lognormal(0.0, 2.0, (n, dim)).astype(np.float32)
g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always", device_log=False)
ctx = _native.context()
ctx.set_option("weighted.rescue", -1)
a, _ = g.minhash_many_arrays(x)
for r in (1, 4, 64):
    ctx.set_option("weighted.rescue", r)
    b, _ = g.minhash_many_arrays(x)
    d = np.argwhere(np.any(a != b, axis=2))
    print("rescue", r, "mismatching (row, sample):", len(d), "of", n * s)
    for row, smp in d[:6]:
        print("   row", row, "sample", smp, "want", a[row, smp], "got", b[row, smp])
ctx.set_option("weighted.rescue", 0)
