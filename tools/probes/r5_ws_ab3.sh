cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for v in abl2 abl34 abl1 abl35; do
  export VITAE_HIP_LIB=build/variants/lib_$v.so
  echo "== $v"
  for shp in "3520 768 3072 fwd" "3520 768 3072 dgrad"; do
    python tools/ws_phase_probe.py $shp 2>/dev/null | grep -E "us/launch|whole loop|B_9" | sed 's/; clocks.*//; s/consumer arrives.*producer;//' | tr '\n' ' '; echo
  done
done
