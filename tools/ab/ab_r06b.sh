mkdir -p gpurun_out/r06b; O=gpurun_out/r06b
timeout 600 python -m pytest tests/test_gpu_convert.py tests/test_gpu_parity.py tests/test_gpu_deferred.py -x -q 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
B="python bench.py --no-cpu-baseline --no-extra-configs"
run() { tag=$1; shift; env MGPU_LIBRARY=libmodes_gpu_exp.so "$@" timeout 200 $B > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['frac'])" 2>/dev/null || tail -3 $O/$tag.log; }
for i in 1 2; do
run old$i MGPU_CONVERT_OLD=1
run lean$i X=1
run side768_$i MGPU_CONV_SIDE=1 MGPU_SLICE_BLOCKS=768
run side768b1024_$i MGPU_CONV_SIDE=1 MGPU_SLICE_BLOCKS=768 MGPU_CONV_SIDE_BLOCKS=1024
run side768b512_$i MGPU_CONV_SIDE=1 MGPU_SLICE_BLOCKS=768 MGPU_CONV_SIDE_BLOCKS=512
run side1024_$i MGPU_CONV_SIDE=1
done
