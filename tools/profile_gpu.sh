#!/bin/bash
# Round profile recipe (run through gpurun): headline bench, rocprofv3 kernel stats, PMC traffic passes.
# usage: tools/profile_gpu.sh <tag> [spp_for_profiles]
TAG=${1:-r1}
PSPP=${2:-128}
PPASS=${3:-128}   # PMC runs: exactly one pass of the headline pass size (the library picks 128 samples per pass at 1080p on an idle 288 GB device), so per-launch numbers are comparable
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench (headline config, unprofiled)"
python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
echo "== rocprofv3 --kernel-trace --stats (spp $PSPP)"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PSPP --cpu-seconds 0 > $OUT/stats_bench.json 2> $OUT/stats.err
echo "== rocprofv3 --pmc FETCH_SIZE (spp $PPASS = one pass)"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PPASS --cpu-seconds 0 > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
echo "== rocprofv3 --pmc TCC_EA0_RDREQ by request size (spp $PPASS = one pass): exact read bytes = 128 a + 64 b + 32 c (profiles/r2_gather_calibration.json)"
timeout -k 5 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/pmc_rd -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PPASS --cpu-seconds 0 > $OUT/pmc_rd_bench.json 2> $OUT/pmc_rd.err
echo "== rocprofv3 --pmc WRITE_SIZE (spp $PPASS = one pass)"
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PPASS --cpu-seconds 0 > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
echo "== rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum (spp $PPASS = one pass)"
timeout -k 5 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PPASS --cpu-seconds 0 > $OUT/pmc_l2_bench.json 2> $OUT/pmc_l2.err
python3 $REPO/tools/summarize_profile.py $OUT $PPASS ${CONFIG:-C2} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged-back payload small: drop the raw traces, keep stats + summaries
find $OUT -name '*kernel_trace.csv' -size +8M -delete
du -sh $OUT
