import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "hlx_env_kernel<608u, 0, false, false"
acc = collections.defaultdict(list)
for r in rows:
    if key in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
