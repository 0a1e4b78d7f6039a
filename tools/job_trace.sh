# kernel trace (not just stats) of one workload: where the blit copies (__amd_rocclr_copyBuffer) sit in a step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/trace; mkdir -p $O
for w in ${WL:-c3 ns}; do
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-check > $O/$w.json 2> $O/$w.err
  f=$(find $O/kt_$w -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $O/${w}_seq.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
out=open(sys.argv[2],'w')
prev=None;cnt=0;acc=0
for r in rows:
    n=r['Kernel_Name'][:48]
    key=(n,r['Grid_Size_X'],r['Stream_Id'])
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if key==prev: cnt+=1; acc+=dur; continue
    if prev: out.write("   x%d total %.1f us\n"%(cnt+1,acc))
    out.write("%9.3f ms %8.1f us  %-48s grid %s wg %s stream %s"%((int(r['Start_Timestamp'])-t0)/1e6,dur,n,r['Grid_Size_X'],r['Workgroup_Size_X'],r['Stream_Id']))
    prev=key;cnt=0;acc=dur
out.write("\n")
PY
  rm -rf $O/kt_$w
done
