# knob sweep of the streaming attention kernels at the patch-8 encoder shape (N = 433, hd 64): forward variant, row blocks per wave
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 0 1 2; do echo "== VITAE_ATTN_FWD=$f"; VITAE_ATTN_FWD=$f python tools/attn_bench.py 2>&1 | grep "N=433\|N=1729"; done
for r in 1 2; do echo "== VITAE_ATTN_RB=$r"; VITAE_ATTN_RB=$r python tools/attn_bench.py 2>&1 | grep "N=433\|N=1729"; done
