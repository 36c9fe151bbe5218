cd $GRAFT_REPO_ROOT
for a in "--no-graph" "--model mae" "--batch 2" "--steps 3 --warmup 1"; do
  echo "== $a"; timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 2 --profile-steps 0 $a 2>&1 | tail -2 | cut -c1-200
done
