cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/libtrace
PYTHONPATH=. rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/libtrace/tr -- python tools/gemm_vs_library.py > gpurun_out/libtrace/run.log 2>&1
f=$(ls gpurun_out/libtrace/tr/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/libtrace/kernel_stats.csv
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/libtrace/tr/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# group consecutive runs of the same kernel name
out=[];cur=None
for r in rows:
    n=r['Kernel_Name']; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if cur and cur[0]==n: cur[1].append(d)
    else:
        cur=[n,[d],r['Grid_Size_X'],r['Workgroup_Size_X'],r.get('LDS_Block_Size',''),r.get('VGPR_Count','')]; out.append(cur)
with open('gpurun_out/libtrace/runs.txt','w') as o:
    for n,ds,g,w,l,v in out:
        if len(ds)>=50:
            ds=sorted(ds); o.write('%4d x median %7.1f us  grid %s wg %s lds %s vgpr %s  %s\n'%(len(ds),ds[len(ds)//2],g,w,l,v,n[:150]))
PY
rm -rf gpurun_out/libtrace/tr
cat gpurun_out/libtrace/runs.txt
