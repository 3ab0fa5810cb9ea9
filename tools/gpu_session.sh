#!/bin/bash
cd /root/repo
out=gpurun_out/r05l; mkdir -p $out
MGPU_LIBRARY=libmodes_gpu_cv2.so timeout 300 python -m pytest tests/test_gpu_convert.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], d.get('stage_ms'))"; }
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/base$i.log 2>/dev/null; summ $out/base$i.log
MGPU_LIBRARY=libmodes_gpu_cv2.so timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/cv2_$i.log 2>/dev/null; summ $out/cv2_$i.log
done
