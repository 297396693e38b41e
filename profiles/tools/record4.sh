#!/bin/bash
# round-4 measurement record: everything DESIGN 5 quotes, from ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4rec; rm -rf $O; mkdir -p $O   # (scratch/ paths below: these tools also live under profiles/tools/)
COMMIT=$1
cd /tmp; export TMPDIR=/tmp
t() { echo "[$(date +%H:%M:%S)] $*"; }
for cfg in "f32 64 64" "bf16 64 64"; do
  set -- $cfg; tag=$1_s$2_b$3
  t "kernel stats $tag"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o ks -- python3 $R/bench.py --dtype $1 --size $2 --batch $3 --serialize --no-cpu --no-secondary --steps 100 --warmup 10 --blocks 1 > $O/ks_$tag.json 2>$O/ks_$tag.err || echo "ks $tag failed"
  cp $(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$tag.csv
  python3 $R/scratch/stats_sum.py $O/kernel_stats_$tag.csv 210 > $O/kernel_stats_$tag.txt
  t "pmc $tag"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$tag -- python3 $R/bench.py --dtype $1 --size $2 --batch $3 --serialize --no-cpu --no-roofline --steps 5 --warmup 2 --blocks 1 > /dev/null 2>$O/pf_$tag.err || echo "fetch $tag failed"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$tag -- python3 $R/bench.py --dtype $1 --size $2 --batch $3 --serialize --no-cpu --no-roofline --steps 5 --warmup 2 --blocks 1 > /dev/null 2>$O/pw_$tag.err || echo "write $tag failed"
  python3 $R/profiles/make_pmc_traffic.py /tmp/pf_$tag /tmp/pw_$tag 7 $COMMIT > $O/pmc_traffic_$tag.json || echo "traffic $tag failed"
  rm -rf /tmp/ks_$tag /tmp/pf_$tag /tmp/pw_$tag
done
t "sq counters"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_sq -- python3 $R/bench.py --serialize --no-cpu --no-roofline --steps 5 --warmup 2 --blocks 1 > /dev/null 2>$O/sq.err || echo "sq failed"
python3 $R/profiles/make_sq_counters.py /tmp/pmc_sq $COMMIT > $O/pmc_sq_counters.json || echo "sq summary failed"
rm -rf /tmp/pmc_sq
cd $R
cp $O/pmc_traffic_f32_s64_b64.json profiles/r04_pmc_traffic.json
cp $O/pmc_traffic_bf16_s64_b64.json profiles/r04_pmc_traffic_bf16_s64_b64.json
t "bench default"
timeout -k 10 300 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 300 $O/bench_default.json
b() { name=$1; shift; t "bench $name"; timeout -k 10 200 python bench.py --no-cpu "$@" > $O/bench_$name.json 2>$O/bench_$name.err || echo "bench $name failed"; }
b f32_s128_b32 --size 128 --latent 128 --batch 32
b f32_s64_b128 --batch 128
b f32_s64_b256 --batch 256
b f32_dist_world1 --dist
b bf16_s64_b64 --dtype bf16
b bf16_s64_b128 --dtype bf16 --batch 128
b bf16_s64_b256 --dtype bf16 --batch 256
b f16_s128_b32 --dtype f16 --size 128 --latent 128 --batch 32
b f16_s128_b64 --dtype f16 --size 128 --latent 128 --batch 64
b mlp_s28_b32 --model mlp --batch 32
t "gloo rehearsal"
SIGGAN_DIST_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench_gpus2_gloo.json 2>$O/bench_gpus2_gloo.err || echo "gloo failed"
t "secondary"
timeout -k 10 300 python profiles/secondary.py --out $O/secondary.json > /dev/null 2>$O/secondary.err || echo "secondary failed"
# (the GPU suite runs in a call of its own: scratch/record4_tests.sh -> parity_margins.json, narrow_parity.json)
t "timeline"
bash scratch/timeline.sh gpurun_out/r4rec
echo all done; ls $O | wc -l
