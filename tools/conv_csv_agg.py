import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r['cfg']!='cfg']
agg=collections.OrderedDict()
for r in rows:
    k=(r['cfg'],r['N'],r['Ho'],r['Wo'],r['cin'],r['cout'],r['k'],r['stride'],r['gx'],r['gy'],r['gz'])
    a=agg.setdefault(k,[0,0.0,0.0])
    a[0]+=1; a[1]+=float(r['us']); a[2]+=float(r['us'])*float(r['tflops'])
tot=sum(a[1] for a in agg.values())
print("total us",tot, "launches",len(rows))
for k,a in sorted(agg.items(), key=lambda x:-x[1][1])[:45]:
    print("cfg%-2s N%s %4sx%-4s cin%-4s cout%-4s k%s s%s grid(%s,%s,%s) n=%-3d us=%9.1f avg=%7.1f TF=%6.1f %5.1f%%"%(k+(a[0],a[1],a[1]/a[0],a[2]/a[1],100*a[1]/tot)))
