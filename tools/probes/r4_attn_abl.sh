cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4attn; mkdir -p $O
for rb in 1 2; do
for v in base abl1 abl2 abl4 abl8 abl16 abl31; do
  L=build/variants/lib_attn_$v.so; [ $v = base ] && L=vit_ae_plus_plus_amd/libvitae_hip.so
  echo "== RB=$rb $v"; VITAE_ATTN_RB=$rb VITAE_HIP_LIB=$L timeout 300 python tools/attn_bench.py 2>/dev/null | grep "N=1729\|N=433"
done; done | tee $O/ablate.txt
