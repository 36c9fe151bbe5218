# Few-column GEMMs of batch 32 on the 128x128 tile at forced splits (and the 256x256 tile unsplit): where does the time go?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for shape in "fwd 3520 768 3072" "fwd 3520 768 768" "fwd 6944 512 2048" "dgrad 3520 768 2304"; do
  for s in 1 2 3 4; do VITAE_BT_TILE=3 VITAE_BT_SPLIT=$s python tools/bt_split_probe.py $shape 2>&1 | tail -1; done
  VITAE_BT_TILE=0 python tools/bt_split_probe.py $shape 2>&1 | tail -1
  VITAE_BT_TILE=-2 python tools/bt_split_probe.py $shape 2>&1 | tail -1
done
} | tee gpurun_out/btsplit.txt
