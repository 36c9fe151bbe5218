for cfg in "3:" "4:0,1,4,8,12" "4:0,1,5,9,12" "5:0,1,3,6,9,12" "3:0,1,6,12" "3:0,2,7,12" "4:0,2,5,8,12" "2:0,1,12"; do
  n=${cfg%%:*}; c=${cfg#*:}
  for rep in 1 2; do
    if [ -n "$c" ]; then r=$(VITAE_ENC_CHUNKS=$n VITAE_ENC_CUTS=$c python bench.py --no-extra --no-cpu-baseline --profile-steps 0 --steps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])");
    else r=$(python bench.py --no-extra --no-cpu-baseline --profile-steps 0 --steps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); fi
    echo "chunks $n cuts [$c] rep $rep: $r ms"
  done
done
