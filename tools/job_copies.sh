# where the blit copies of a step come from: kernel trace + memory-copy trace of one workload (WL), copies over 50 us listed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/copies; mkdir -p $O
for w in ${WL:-c3 ns}; do
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-check > $O/$w.json 2> $O/$w.err
  ls $O/kt_$w/*/ | head
  f=$(find $O/kt_$w -name "*memory_copy_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows),"copies; columns",list(rows[0].keys()) if rows else None)
big=[r for r in rows if (int(r['End_Timestamp'])-int(r['Start_Timestamp']))>50000]
for r in big[-40:]:
    print({k:r[k] for k in r if k not in ('Correlation_Id',)}, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us')
PY
  rm -rf $O/kt_$w
done
