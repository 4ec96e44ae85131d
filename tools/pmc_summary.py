import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
disp=collections.OrderedDict()
for r in rows:
    d=disp.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'],'grid':int(r['Grid_Size']),'dur':(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3})
    d[r['Counter_Name']]=d.get(r['Counter_Name'],0)+float(r['Counter_Value'])
items=list(disp.values())
idx=[i for i,d in enumerate(items) if 'igemm_kernel<4, 1, true>' in d['name'] or 'smallc' in d['name']]
for d in items[idx[-1]:]:
    if 'igemm2' not in d['name'] or d['dur']<float(sys.argv[2]) if len(sys.argv)>2 else 150: continue
    cyc=d.get('GRBM_GUI_ACTIVE',0)/8
    mf=d.get('SQ_INSTS_MFMA',1)
    print('%-28s dur=%7.1fus mfma_busy=%4.0f%% per-MFMA: valu=%.2f salu=%.2f lds=%.2f'%(d['name'].replace('alq::','').replace('void ','')[:28],d['dur'],100*d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(cyc*1024),(d.get('SQ_INSTS_VALU',0)-mf)/mf,d.get('SQ_INSTS_SALU',0)/mf,d.get('SQ_INSTS_LDS',0)/mf))
