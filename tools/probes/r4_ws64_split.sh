# ws64 (tile 5) at forced splits on the long-reduction shapes of the batch-4 / batch-8 step, against the 64-row family's own split
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for shape in "fwd 440 768 3072" "dgrad 440 768 2304" "dgrad 440 768 3072" "fwd 868 512 2048" "dgrad 868 512 2048" "fwd 880 768 3072" "dgrad 1736 512 2048"; do
  for s in 1 2 3 4 6; do VITAE_BT_TILE=5 VITAE_BT_SPLIT=$s python tools/bt_split_probe.py $shape 2>&1 | tail -1; done
  VITAE_BT_TILE=-2 python tools/bt_split_probe.py $shape 2>&1 | tail -1
done
} | tee gpurun_out/ws64split.txt
