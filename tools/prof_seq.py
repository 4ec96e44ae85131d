import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'igemm_kernel<4, 1, true>' in r['Kernel_Name'] or 'direct_conv' in r['Kernel_Name']]
s=idx[len(idx)//2]; e=idx[len(idx)//2+1]
tot=0; gem=0
for r in rows[s:e]:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('alq::','').replace('void ','').split('(')[0][:34]
    tot+=en-st
    if 'igemm' in name: gem+=en-st
    if len(sys.argv)>2 or 'igemm' in name:
        print('%-36s grid=%6d vgpr=%s dur=%8.1f us'%(name,int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),r['VGPR_Count'],(en-st)/1e3))
print('batch: kernels %.1f us, gemm %.1f us, wall %.1f us'%(tot/1e3,gem/1e3,(int(rows[e]['Start_Timestamp'])-int(rows[s]['Start_Timestamp']))/1e3))
