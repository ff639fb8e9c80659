import numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, '.')
from datasketch_amd import WeightedMinHashGenerator
from oracle import oracle as O
def check(dim, s, x, name):
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    out, ne = g.minhash_many_arrays(x)
    csr = sp.csr_matrix(x); csr.sort_indices()
    want, wn = O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)
    ok = np.array_equal(out[wn.astype(bool)], want[wn.astype(bool)]) and np.array_equal(ne, wn)
    print(name, "OK" if ok else "MISMATCH", flush=True)
    if not ok:
        bad = np.argwhere(np.any(out != want, axis=2) & wn.astype(bool)[:, None])
        print(" ne", ne.tolist()[:16], "want", wn.tolist()[:16])
        print(" first mismatches (row, sample):", bad[:10].tolist())
        for r, i in bad[:4]:
            print("  got", out[r, i].tolist(), "want", want[r, i].tolist())
x = np.array([[1, 0, 3, 0, 0.5, 2, 0, 7], [0] * 8, [2] * 8], dtype=np.float32)
check(8, 4, x, "small 8x4")
rng = np.random.RandomState(0)
for dim, s, n in [(8, 4, 3), (8, 64, 8), (16, 64, 8), (64, 32, 30), (5, 3, 9), (7, 70, 17), (300, 100, 48), (4096, 128, 16), (513, 128, 33)]:
    x = rng.uniform(0, 100, (n, dim)).astype(np.float32)
    check(dim, s, x, f"dense {n}x{dim} s={s}")
    x[rng.random_sample(x.shape) < 0.5] = 0
    check(dim, s, x, f"half  {n}x{dim} s={s}")
