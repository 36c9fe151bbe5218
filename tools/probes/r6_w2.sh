cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
for w in "dec.fc1" "decoder" "decoder,enc.fc1" "decoder,enc.fc2" "decoder,enc.proj" "decoder,enc.qkv" "dec.fc1" "decoder"; do
VITAE_W2="$w" python tools/w2_parity.py 2>/dev/null | tail -1
done | tee $O/w2_parity2.txt
