cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in "dec.fc1" "decoder,enc.proj" "decoder"; do echo "W2=$w"; VITAE_W2="$w" timeout 900 python -m pytest tests/test_gpu_model.py -x -q -s -m gpu -k "(patch8 or config5 or config4_vs) and bf16" 2>&1 | grep "worst loss error\|passed\|failed"; done
