"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name."""
import csv, glob, os, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if "corr" not in k and "warp" not in k and "ternary" not in k:
                continue
            m = re.search(r"(\w+_kernel)", k)          # the kernel's base name (template arguments and launches pooled)
            acc[m.group(1) if m else k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print("   %-40s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
