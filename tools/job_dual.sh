export AB=r8 WL="c1 ns"
bash tools/job_ab.sh
