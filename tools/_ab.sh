mkdir -p gpurun_out/r04as
L=gpurun_out/r04as/slots.log
for rep in 1 2 3; do
for cfg in "24 10 58" "32 10 74" "32 12 76"; do
  set -- $cfg
  echo "== slots $1 workers $2 depth $3" >> $L
  timeout -k 10 150 python3 tools/acc_probe.py --grid naf:0 --no-dump --proofs 4 --stream 3000 --repeat 2 --gpu-slots $1 --witness-workers $2 --depth $3 2>&1 | tail -1 | sed 's/.*same_proof.: True, //' >> $L
done
done
