mkdir -p gpurun_out/r06e; O=gpurun_out/r06e
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-extra-configs "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; g=s['convert']+s['sweep']+s['slice']+s['prescreen']; n=d['config']['samples_per_feed']; print('$tag', d['value'], d.get('ms_per_feed'), 'gpu-sum/537M', round(g*536870912/n,3), s, d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2; do
run c1024_$i
run c640_$i --chunk-buffers 640 --samples $((2560*131072))
run c768_$i --chunk-buffers 768 --samples $((3072*131072))
run c896_$i --chunk-buffers 896 --samples $((3584*131072))
run c1152_$i --chunk-buffers 1152 --samples $((4608*131072))
done
