# the N > 1 machinery at a world of one rank (VITAE_FORCE_DDP=1): both exchange routes' step times under a branch-fork setting
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 VITAE_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d["config"]; e=c.get("exchange") or {}; a=c.get("also_exchange") or {}; print("reported", d["ms_per_step"], "|", str(e.get("route"))[:28], e.get("ms_per_step"), "|", str(a.get("route"))[:28], a.get("ms_per_step"), a.get("error"))' 2>&1 | tail -1)"; }
for i in 1 2; do run X=0; done
