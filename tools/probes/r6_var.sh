cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10; do python tools/step_variance.py 2>/dev/null | tail -1; done
