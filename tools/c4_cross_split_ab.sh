Q="--workload c4 --steps 20 --warmup 4 --no-cpu-baseline --no-torch-gpu-baseline --no-live-traffic --no-second-precision --no-io-rates --no-batch1 --no-configs"
val() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config']['cross_attention_key_split'])"; }
for rep in 1 2; do
  echo "c4 keys split by the batch:   3 in flight $(python3 bench.py $Q 2>/dev/null | val) | one stream $(python3 bench.py $Q --inflight 1 2>/dev/null | val)"
  echo "c4 default (no split):        3 in flight $(python3 bench.py $Q --no-cross-ksplit-auto 2>/dev/null | val) | one stream $(python3 bench.py $Q --inflight 1 --no-cross-ksplit-auto 2>/dev/null | val)"
done
