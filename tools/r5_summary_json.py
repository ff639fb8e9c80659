#!/usr/bin/env python3
"""tools/r5_summary_json.py <gpurun_out/r5_<tag>> -> JSON on stdout (profiles/r05_traffic_lsh_sort.json).

Per kernel of tools/r5_probe.py (bucketing passes, digest / pack / fused kernels, the merge calibration): duration from the
kernel trace, L2 <-> fabric bytes from the TCC_EA0 request counters (own --pmc passes; byte formula and calibration as
tools/traffic_summary.py), L2 hit share, LDS bank-conflict share, the share of wave cycles spent waiting on an instruction,
VALU issue share -- next to the algorithmic bytes of the launch."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

N, BANDS, K = 1_250_000, 32, 256
ALG = {  # algorithmic bytes per launch (SURVEY.md section 8d / BASELINE.md section 4)
    "lsh_bin_scatter_kernel<Digest64BM": N * BANDS * (8 + 12),   # digest in, (digest, row) out
    "lsh_bin_scatter_kernel<Digest64,": N * BANDS * (8 + 12),
    "digests_to_band_major_kernel": N * BANDS * 16,
    "lsh_bin_sort_kernel": N * BANDS * (12 + 12),                 # (digest, row) in and out
    "bbit_digest_fused_kernel": N * (4 * K + K // 8 + 8 * BANDS),
    "band_digest_kernel": N * (4 * K + 8 * BANDS),
    "band_digest_bm_kernel": N * (4 * K + 8 * BANDS),
    "bbit1_wide_kernel": N * (4 * K + K // 8),
    "minhash_merge_kernel": 3 * 128_000_000 * 8,
}


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("mhx::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def counters(db):
    con = sqlite3.connect(db)
    per = defaultdict(lambda: defaultdict(float))
    for kernel, disp, counter, value in con.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        per[(short(kernel), counter)][disp] += value
    out = defaultdict(dict)
    for (kernel, counter), d in per.items():
        out[kernel][counter] = sum(d.values()) / len(d)
    return out


def main():
    root = sys.argv[1]
    merged = defaultdict(dict)
    for db in glob.glob(os.path.join(root, "*", "*.db")):
        if os.path.basename(os.path.dirname(db)) == "trace":
            con = sqlite3.connect(db)
            for name, calls, avg, mn in con.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
                merged[short(name)].update({"dispatches": calls, "avg_us": avg / 1e3, "min_us": mn / 1e3})
        else:
            for kernel, c in counters(db).items():
                merged[kernel].update(c)
    out = {}
    for kernel, c in sorted(merged.items()):
        if "avg_us" not in c or not any(key in kernel for key in ("lsh_bin", "digest", "bbit", "merge")):
            continue
        rec = {"avg_us": round(c["avg_us"], 1), "min_us": round(c["min_us"], 1), "dispatches": c["dispatches"]}
        if "TCC_EA0_RDREQ_sum" in c:
            n, n32, n64 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0)
            rec["read_bytes"] = int(32 * n32 + 64 * n64 + 128 * (n - n32 - n64))
        if "TCC_EA0_WRREQ_sum" in c:
            n, n64 = c["TCC_EA0_WRREQ_sum"], c.get("TCC_EA0_WRREQ_64B_sum", 0)
            rec["write_bytes"] = int(64 * n64 + 32 * (n - n64))
            rec["write_requests_of_32_bytes_share"] = round((n - n64) / n, 3) if n else None
        alg = next((v for key, v in ALG.items() if kernel.startswith(key)), None)
        if alg:
            rec["algorithmic_bytes"] = alg
            rec["algorithmic_GBps"] = round(alg / (c["avg_us"] * 1e-6) / 1e9, 1)
            rec["frac_of_8_TBps"] = round(alg / (c["avg_us"] * 1e-6) / 8e12, 3)
        if "read_bytes" in rec and "write_bytes" in rec:
            rec["traffic_bytes"] = rec["read_bytes"] + rec["write_bytes"]
            rec["traffic_GBps"] = round(rec["traffic_bytes"] / (c["avg_us"] * 1e-6) / 1e9, 1)
            if alg:
                rec["traffic_over_algorithmic"] = round(rec["traffic_bytes"] / alg, 2)
        if c.get("TCC_REQ_sum"):
            rec["l2_hit_share"] = round(c.get("TCC_HIT_sum", 0) / c["TCC_REQ_sum"], 3)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_share_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        if c.get("SQ_WAVE_CYCLES"):
            rec["wait_inst_share_of_wave_cycles"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
            if c.get("GRBM_GUI_ACTIVE"):
                rec["mean_waves_per_cu"] = round(4 * c["SQ_WAVE_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8) / 256, 1)  # SQ_WAVE_CYCLES counts quads of cycles
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_INSTS_VALU"):
            rec["valu_issue_frac"] = round(4 * c["SQ_INSTS_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
        out[kernel] = rec
    print(json.dumps({
        "workload": f"tools/r5_probe.py: {N} x {K} uint32 signatures, {BANDS} bands x 8 (config 3 / 5's per-GPU shard)",
        "source": "rocprofv3 --kernel-trace and separate --pmc passes (tools/r5_passes.sh); counters are means per dispatch, summed over instances",
        "byte_formula": "reads 32*RDREQ_32B + 64*RDREQ_64B + 128*(RDREQ - RDREQ_32B - RDREQ_64B); writes 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B); "
                        "minhash_merge_kernel (2.048 GB read, 1.024 GB written by construction) is the calibration",
        "kernels": out,
    }, indent=1))


if __name__ == "__main__":
    main()
