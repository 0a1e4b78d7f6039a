# generic GPU-box runner: tools/r2_run.sh <outdir-name> <<< "commands"   (commands read from the file given as $2)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/$1; mkdir -p $O
bash "$2"
