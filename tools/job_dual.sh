export AB=prio2 WL="c1 ns"
bash tools/job_ab.sh
