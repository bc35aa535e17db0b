# bench.py over leaf_refine modes (include/rt_hip.h); RT_HIP_LIB may name a variant library
mkdir -p gpurun_out
OWN=1073741824
for r in ${SWEEP:-0 32 $((OWN+24)) $((OWN+40))}; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 24 --leaf-refine $r | grep "^{" > gpurun_out/rm_$r.json || exit 1
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('ms_per_frame_one_in_flight'), d['roofline']['wave_box_fetches_per_ray'], d['roofline']['wave_record_fetches_per_ray'], d['config'].get('scene_build_s'))" gpurun_out/rm_$r.json
done
