# modelled ring (design aid): one vs two decoder buckets
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 60 --profile-steps 0 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['grad_allreduce'], d['config']['final_losses'][0])"; }
export VITAE_FORCE_DDP=1
for dc in 1 2; do
  echo "dec_chunks $dc, no model: $(VITAE_DEC_CHUNKS=$dc run)"
  for bw in 300 200 150; do echo "dec_chunks $dc busbw $bw: $(VITAE_DEC_CHUNKS=$dc VITAE_DDP_SIM_BUSBW=$bw run)"; done
done
