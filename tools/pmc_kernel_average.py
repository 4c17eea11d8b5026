import csv,glob,collections,sys
for d in sys.argv[1:]:
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sym" in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {c: round(sum(x)/len(x),1) for c,x in acc.items()})
