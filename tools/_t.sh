mkdir -p gpurun_out/r04ae
L=gpurun_out/r04ae/c15.log
for m in 2 1 3 2; do
  echo "== mult $m" >> $L
  ZKHIP_STREAM_SLICE_MULT=$m timeout -k 10 200 python3 tools/acc_probe.py --grid naf:0 --no-dump --proofs 4 --prove-stream 2000 --repeat 2 >> $L 2>&1 || exit 1
done
echo "== zeth" >> $L
timeout -k 10 300 python3 tools/acc_probe.py --grid naf:0 --no-dump --proofs 4 --prove-stream 1000 --repeat 2 --nested-inputs 9 >> $L 2>&1 || exit 1
timeout -k 10 600 python3 -m pytest tests/test_aggregator_gpu.py tests/test_prover_gpu.py tests/test_multi_device_gpu.py -m gpu -x -q >> $L 2>&1
