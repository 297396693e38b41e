#!/bin/bash
# copy gpurun_out/r4rec (scratch/record.sh) into the tracked profiles/r04_* files
cd /root/repo; O=gpurun_out/r4rec
for f in $O/bench_*.json; do n=$(basename $f .json); python - "$f" "profiles/r04_${n}.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        json.dump(json.loads(l), open(sys.argv[2],'w'), indent=1); break
PY
done
mv profiles/r04_bench_gpus2_gloo.json profiles/r04_bench_gpus2_gloo_rehearsal.json
python - <<'PY'
import json
p='/root/repo/profiles/r04_bench_gpus2_gloo_rehearsal.json'
d=json.load(open(p)); d["note"]="REHEARSAL of `python bench.py --gpus 2` (no launcher around it) on the ONE-GPU box: SIGGAN_DIST_BACKEND=gloo, both ranks share cuda:0, buckets reduced by torch.distributed between the step halves. It shows the self-launch path end to end (n_gpus 2, global_batch 128, one JSON line); its images/s is two ranks time-slicing one GPU, not a scaling number."
json.dump(d,open(p,'w'),indent=1)
for l in open('/root/repo/gpurun_out/r4rec/ks_f32_s64_b64.json'):
    if l.startswith('{'): json.dump(json.loads(l), open('/root/repo/profiles/r04_bench_serialized_under_rocprof.json','w'), indent=1)
PY
cp $O/kernel_stats_f32_s64_b64.csv profiles/r04_kernel_stats_serialized.csv; cp $O/kernel_stats_bf16_s64_b64.csv profiles/r04_kernel_stats_serialized_bf16.csv
cp $O/kernel_stats_f32_s64_b64.txt profiles/r04_kernel_stats_per_step.txt; cp $O/kernel_stats_bf16_s64_b64.txt profiles/r04_kernel_stats_per_step_bf16.txt
cp $O/pmc_traffic_f32_s64_b64.json profiles/r04_pmc_traffic.json; cp $O/pmc_traffic_bf16_s64_b64.json profiles/r04_pmc_traffic_bf16_s64_b64.json; cp $O/pmc_sq_counters.json profiles/r04_pmc_sq_counters.json
cp $O/secondary.json profiles/r04_secondary.json; cp $O/parity_margins.json profiles/r04_parity_margins.json; cp $O/narrow_parity.json profiles/r04_narrow_parity.json; cp $O/timeline.txt profiles/r04_timeline_pipelined_step.txt
