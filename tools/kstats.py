import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("k_bin","k_scan","k_hist","k_scatter","k_accum")): print(n[:40], r["Calls"], r["AverageNs"])
