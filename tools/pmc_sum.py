import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    acc[(r["Kernel_Name"][:48],r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v), max(v))
