import csv, collections, glob, sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp']); r['n']=r['Kernel_Name'].split('(')[0]
rows.sort(key=lambda r:r['s'])
idx=[i for i,r in enumerate(rows) if r['n'].startswith('k_intra')]
seg=rows[idx[-1]:]
t0=seg[0]['s']
print('pass span us', (max(r['e'] for r in seg)-t0)/1e3)
me=[r for r in seg if r['n'].startswith('k_me_pool')]
inter=[r for r in seg if r['n'].startswith('k_inter_fused')]
gaps=[(me[i+1]['s']-inter[i]['e'])/1e3 for i in range(len(me)-1)]
print('gap inter->next me: mean %.1f min %.1f max %.1f'%(sum(gaps)/len(gaps),min(gaps),max(gaps)))
per=[(me[i+1]['s']-me[i]['s'])/1e3 for i in range(len(me)-1)]
print('step period mean %.1f'%(sum(per)/len(per)))
