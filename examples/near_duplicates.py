#!/usr/bin/env python3
"""Near-duplicate detection, the datasketch way, with the whole chain in bulk.

    python examples/near_duplicates.py [--docs 20000] [--gpu-mode detect]

1. ``MinHash.bulk_signatures``  -- one signature row per document (byte tokens, default SHA-1 hash:
   packed by csrc/pack_module.c, hashed and permuted on the device);
2. ``lsh_bulk.candidate_pairs`` -- the pairs a ``MinHashLSH(params=(b, r))`` would report for each other
   (band digests, one radix sort, run detection, sort + unique on the device);
3. ``lsh_bulk.jaccard_pairs``   -- ``MinHash.jaccard`` of every candidate pair, then a threshold.

The same objects still drop into the reference's own index: the last lines insert a few rows into a
``datasketch.MinHashLSH`` (if that package is importable) through ``lsh_bulk.insert_bulk`` and query it.
With ``--gpu-mode disable`` everything runs on the numpy paths and gives the same answer.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datasketch_amd import MinHash  # noqa: E402
from datasketch_amd import lsh_bulk  # noqa: E402


def synthetic_corpus(n_docs: int, seed: int = 0):
    """Documents of 40..120 word tokens; every tenth document is an edited copy of an earlier one."""
    rng = np.random.RandomState(seed)
    vocab = [b"w%d" % i for i in range(50_000)]
    docs, truth = [], []
    for i in range(n_docs):
        if i % 10 == 9:
            src = int(rng.randint(0, i))
            words = list(docs[src])
            for pos in rng.randint(0, len(words), max(1, len(words) // 20)):  # edit ~5 % of the words
                words[pos] = vocab[rng.randint(0, len(vocab))]
            truth.append((src, i))
        else:
            words = [vocab[j] for j in rng.randint(0, len(vocab), rng.randint(40, 121))]
        docs.append(words)
    return docs, truth


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=20_000)
    ap.add_argument("--num-perm", type=int, default=128)
    ap.add_argument("--bands", type=int, default=32)
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--threshold", type=float, default=0.7)
    ap.add_argument("--gpu-mode", default="detect", choices=["detect", "always", "disable"])
    args = ap.parse_args(argv)

    docs, truth = synthetic_corpus(args.docs)
    t0 = time.perf_counter()
    sig = MinHash.bulk_signatures(docs, num_perm=args.num_perm, seed=1, gpu_mode=args.gpu_mode)
    t1 = time.perf_counter()
    pairs = lsh_bulk.candidate_pairs(sig, args.bands, args.rows, gpu_mode=args.gpu_mode)
    t2 = time.perf_counter()
    est = lsh_bulk.jaccard_pairs(sig, pairs, gpu_mode=args.gpu_mode)
    keep = pairs[est >= args.threshold]
    t3 = time.perf_counter()
    found = set(map(tuple, keep.tolist()))
    recall = sum((a, b) in found for a, b in truth) / max(1, len(truth))
    print(f"{len(docs)} documents -> signatures {t1 - t0:.3f} s, {len(pairs)} candidate pairs {t2 - t1:.3f} s, "
          f"{len(keep)} pairs with estimated Jaccard >= {args.threshold} in {t3 - t2:.3f} s; "
          f"{recall:.1%} of the {len(truth)} planted near-duplicates found")
    try:  # the reference's index still works on these rows
        from datasketch import MinHashLSH
    except ImportError:
        MinHashLSH = None
    if MinHashLSH is not None:
        lsh = MinHashLSH(num_perm=args.num_perm, params=(args.bands, args.rows))
        lsh_bulk.insert_bulk(lsh, range(min(2000, len(docs))), sig[:2000], gpu_mode=args.gpu_mode)
        print("datasketch.MinHashLSH.query(doc 0) ->", sorted(lsh.query(MinHash(seed=1, hashvalues=sig[0])))[:5])
    return {"signatures": sig, "pairs": pairs, "kept": keep, "recall": recall}


if __name__ == "__main__":
    main()
