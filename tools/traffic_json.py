"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.
FETCH_SIZE is doubled (gfx950 tallies a 128-byte request of a wide coalesced read at 64 bytes,
MI355X_MICROARCH.md section HBM); both counters are in KB."""
import collections, csv, glob, hashlib, json, os, sys
O, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "pmc_corr_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k, c = row["Kernel_Name"], row["Counter_Name"]
        if c not in ("FETCH_SIZE", "WRITE_SIZE") or not ("corr_fwd2_kernel" in k or "corr_mfma_bwd4_kernel" in k):
            continue
        key = "fwd" if "fwd2" in k else ("bwd_wrt1" if ("Lb1E" in k or "<true>" in k) else "bwd_wrt0")
        acc[key][c].append(float(row["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def hbm(k):
    return (2 * mean[k]["FETCH_SIZE"] + mean[k]["WRITE_SIZE"]) * 1024


sys.path.insert(0, ROOT)
from bench import correlation_kernel_source_sha16  # noqa: E402  (the same hash bench.py checks the record against)
B, C, H8, W8, D = 32, 256, 48, 160, 441
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) on tools/corr_only.py, "
              "B=32 [32,256,48,160]; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); KB -> bytes x1024",
    "kernel_source_sha16": correlation_kernel_source_sha16(),
    "per_launch_kb": mean,
    "corr_fwd": {"hbm_bytes": hbm("fwd"), "algorithmic_bytes": 4.0 * B * H8 * W8 * (2 * C + D)},
    "corr_bwd_both_launches": {"hbm_bytes": hbm("bwd_wrt0") + hbm("bwd_wrt1"), "algorithmic_bytes": 4.0 * B * H8 * W8 * (D + 4 * C),
                               "note": "topdiff is read once per gradient launch, the algorithmic count has it once"},
}
for k in ("corr_fwd", "corr_bwd_both_launches"):
    out[k]["ratio"] = out[k]["hbm_bytes"] / out[k]["algorithmic_bytes"]
print(json.dumps(out, indent=1))
