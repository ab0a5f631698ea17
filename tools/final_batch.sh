#!/bin/bash
# Measurement batch for profiles/: run on the GPU box from the repo root:  bash tools/final_batch.sh <tag> [a|b]
# (two halves, each inside one gpurun call's time limit: a = PMC traffic, kernel stats, SQ counters, tool matrix, the default line;
#  b = timelines, slice-band and two-rank runs, the round's same-box A/B runs, the ingest split)
# PMC passes are separate rocprofv3 runs (counters never together with --stats); the program itself follows "--".
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd $R
Q="--no-extras --cpu-frames 0"
PART=${2:-ab}
if [[ $PART == *a* ]]; then
cd /tmp; export TMPDIR=/tmp
for wl in 1080p 2160p; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 0 $Q --no-profile --no-check-pass > $O/pmcf_$wl.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 0 $Q --no-profile --no-check-pass > $O/pmcw_$wl.log 2>&1
  echo "pmc $wl done"
done
cd $R
for wl in 1080p 2160p; do python3 tools/pmc_summary.py $O/pmcf_$wl $O/pmcw_$wl $wl > /dev/null; done
cp profiles/pmc_traffic.json $O/
cd /tmp
for wl in 1080p 2160p; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 $R/bench.py --workload $wl --steps 6 --warmup 3 $Q --no-profile --no-check-pass > $O/stats_$wl.log 2>&1
  echo "stats $wl done"
done
# every preset tool on, kernels alone on the chip
VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tools -- python3 $R/bench.py --steps 2 --warmup 1 $Q --no-profile --subpel 2 --deblock --crf 26 --intra-modes 3 > $O/stats_tools.log 2>&1
# SQ counters, every kernel alone on the chip (one stream)
for wl in 1080p 2160p; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/sq_$wl/${c%% *} -- python3 $R/bench.py --workload $wl --steps 3 --warmup 0 $Q --no-profile --no-check-pass > $O/sq_$wl.log 2>&1
  done
  echo "sq $wl done"
done
cd $R
for wl in 1080p 2160p; do python3 tools/sq_summary.py $O/sq_$wl $wl profiles/sq_counters.json > $O/sq_counters_${wl}_isolated.txt 2> $O/sq_summary_$wl.err; done
cp profiles/sq_counters.json $O/
timeout -k 10 300 python bench.py --workload 2160p $Q > $O/bench_2160p.json 2> $O/bench_2160p.err; cut -c1-200 $O/bench_2160p.json
# the optional coding tools (each checked bit-exact against the oracle on the first GOP)
for v in "crf:--crf 26" "subpel2:--subpel 2" "deblock:--deblock" "deblock_gop_batch10:--deblock --gop-batch 10" "intra4x4:--intra-modes 3" "all_tools:--crf 26 --intra-modes 3 --subpel 2 --deblock" "all_tools_gop_batch10:--crf 26 --intra-modes 3 --subpel 2 --deblock --gop-batch 10"; do
  timeout -k 10 300 python bench.py --no-extras --cpu-frames 30 ${v#*:} > $O/bench_tool_${v%%:*}.json 2> $O/bench_tool_${v%%:*}.err; cut -c1-160 $O/bench_tool_${v%%:*}.json
done
# the default line last in this half: it quotes the PMC traffic and the SQ counters collected above (same device sources)
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-240 $O/bench_default.json
fi
if [[ $PART == *b* ]]; then
bash tools/timeline.sh ${1:-final}/tl1080 1080p 9 > /dev/null 2>&1; cp $O/tl1080/timeline_1080p.txt $O/ 2>/dev/null
bash tools/timeline.sh ${1:-final}/tl2160 2160p 9 > /dev/null 2>&1; cp $O/tl2160/timeline_2160p.txt $O/ 2>/dev/null
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --mode slices --workload 2160p $Q > $O/bench_slices1.json 2> $O/bench_slices1.err; cut -c1-200 $O/bench_slices1.json
# two ranks on the one GPU (gloo rehearsal of the N > 1 run: bench.py starts its own ranks; the line carries the config-5 leg as slices_2160p)
timeout -k 10 500 python bench.py --gpus 2 --device 0 --backend gloo --steps 3 --warmup 1 $Q --slices-steps 2 > $O/bench_streams2.json 2> $O/bench_streams2.err
cut -c1-160 $O/bench_streams2.json
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --steps 3 --warmup 1 --gpus 2 --backend gloo --device 0 $Q --slices-steps 2 > $O/bench_streams2_torchrun.json 2> $O/bench_streams2_torchrun.err
grep -o '"equals_one_rank_bitstream": [a-z]*' $O/bench_streams2.json $O/bench_streams2_torchrun.json
# the round's same-box A/B runs (medians of three rounds in rotation): the launcher's knobs, then the GOPs in lock-step
WL=1080p REPS=3 STEPS=30 bash tools/ab_env.sh ${1:-final}/ab_1080 "VENC_X=0" "VENC_HOST_THREADS=2" "VENC_NUMA_BIND=0" "VENC_ENTROPY_CUS=0" "VENC_ENTROPY_CUS=192" "VENC_ME_XCD=1" "VENC_CHAINS=2" > $O/ab_launcher_1080p.txt 2>&1
for gb in 10 20 30 40; do AB_FLAGS="--gop-batch $gb" WL=1080p REPS=3 STEPS=30 bash tools/ab_env.sh ${1:-final}/ab_1080_gb$gb "GOPS_IN_LOCK_STEP=$gb" 2>&1 | grep median >> $O/ab_launcher_1080p.txt; done
WL=2160p STEPS=12 REPS=3 bash tools/ab_env.sh ${1:-final}/ab_2160 "VENC_X=0" "VENC_ME_XCD=0" "VENC_ENTROPY_CUS=0" "VENC_ENTROPY_CUS=160" > $O/ab_launcher_2160p.txt 2>&1
for gb in 10 15 20; do AB_FLAGS="--gop-batch $gb" WL=2160p REPS=3 STEPS=12 bash tools/ab_env.sh ${1:-final}/ab_2160_gb$gb "GOPS_IN_LOCK_STEP=$gb" 2>&1 | grep median >> $O/ab_launcher_2160p.txt; done
cat $O/ab_launcher_1080p.txt $O/ab_launcher_2160p.txt
[ -f video-encoder_amd/csrc/libvenc_hip_stamps.so ] && { VENC_HIP_LIB=libvenc_hip_stamps.so VENC_NO_OVERLAP=1 python tools/ent_stamps.py 1080p 2>/dev/null | grep -v amdgpu.ids > $O/entropy_block_lifetimes_1080p.txt; cat $O/entropy_block_lifetimes_1080p.txt; }
./tools/h2d_ubench 2>/dev/null | grep "depth [12]" > $O/h2d_ubench.txt; HSA_ENABLE_SDMA=0 ./tools/h2d_ubench 2>/dev/null | grep "depth [12]" | sed 's/^/blit kernels (HSA_ENABLE_SDMA=0): /' >> $O/h2d_ubench.txt; cat $O/h2d_ubench.txt
python tools/ingest_trace.py 2>&1 | grep -v "venc trace:  \|amdgpu.ids" > $O/ingest_split.txt; cat $O/ingest_split.txt
fi
for d in $O/pmcf_* $O/pmcw_* $O/stats_* $O/sq_1080p $O/sq_2160p $O/tl1080 $O/tl2160; do [ -d "$d" ] && find $d -name "*agent_info.csv" -delete; done
du -sh $O
