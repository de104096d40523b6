import csv, collections, glob, sys
for run in sys.argv[1:]:
    for f in glob.glob(f"{run}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            if "hwer" in k:
                print(run.split("/")[-1], k, {c: "%.3g" % (sum(v)/len(v)) for c, v in sorted(d.items())})
