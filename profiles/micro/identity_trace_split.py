import csv,collections,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
per=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    m=re.search(r"(id_\w+?)(<\d>)?\(", n)
    if m:
        per[m.group(0).rstrip("(")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=[0,0]
for k,v in per.items():
    h=len(v)//2
    a,b=sum(v[2:h])/max(1,h-2), sum(v[h+2:])/max(1,len(v)-h-2)
    mult = 2 if "scan" in k else 1
    tot[0]+=a*mult; tot[1]+=b*mult
    print("  %-24s random %7.1f  wavefront %7.1f" % (k, a, b))
print("  sum                      random %7.1f  wavefront %7.1f" % tuple(tot))
