import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
total = "--sum" in sys.argv          # --sum: the counter summed over a kernel's launches instead of averaged
for k, d in agg.items():
    if "lmono" in k:
        print(k, {c: round(v if total else v / cnt[(k, c)]) for c, v in d.items()})
