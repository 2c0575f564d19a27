import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: sys.exit("no counter csv under " + sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:48]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k, {c: (round(x / n[(k, c)]), n[(k, c)]) for c, x in v.items()})
