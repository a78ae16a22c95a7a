python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --no-live-traffic --no-torch-gpu-baseline 2>&1 | tail -1 > gpurun_out/fold2_bench.json
python -c "
import json,sys; d=json.load(open('gpurun_out/fold2_bench.json')); print(d['value'], d['ms_per_step'], d['second_precision']['value'], d['parity'], d['second_precision']['parity'], d['roofline']['achieved'], d['config']['flops_per_image'])"
