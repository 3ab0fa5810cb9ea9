mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py -x -q 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt | cut -c1-300
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-6000
timeout 600 python bench.py --chunk-buffers 0 --ahead 1 --no-cpu-baseline > $O/bench_old.log 2>&1; tail -1 $O/bench_old.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('old protocol', d['value'], d['ms_per_feed'], {k:(v['msamples_s'], v['msamples_s_both_repetitions']) for k,v in d['configs'].items()})"
