cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for e in "VITAE_TARGET_FORK=start" "VITAE_TARGET_FORK=embed" "VITAE_TARGET_FORK=decoder" "VITAE_TARGET_FORK=start" "VITAE_TARGET_FORK=embed" "VITAE_TARGET_FORK=decoder"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
