#!/bin/bash
# Measurement batch for profiles/: run on the GPU box from the repo root:  bash tools/final_batch.sh <tag> [a|b]
# (two halves, each inside one gpurun call's time limit: a = PMC traffic, bench lines, tool matrix, kernel stats, SQ counters;
#  b = slice-band and two-rank runs, phase budgets, microbenchmarks, the round's same-box A/B runs)
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
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$wl -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 $Q --no-profile > $O/pmcf_$wl.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$wl -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 $Q --no-profile > $O/pmcw_$wl.log 2>&1
  echo "pmc $wl done"
done
cd $R
for wl in 1080p 2160p; do python3 tools/pmc_summary.py $O/pmcf_$wl $O/pmcw_$wl $wl > /dev/null; done
cp profiles/pmc_traffic.json $O/
timeout -k 10 300 python bench.py --workload 2160p $Q > $O/bench_2160p.json 2> $O/bench_2160p.err; cut -c1-240 $O/bench_2160p.json
timeout -k 10 300 python bench.py --serial-steps $Q > $O/bench_1080p_serial.json 2> $O/bench_1080p_serial.err
# the optional coding tools (each checked bit-exact against the oracle on the first GOP), 1080p and for sub-sample refinement 2160p
for v in "crf:--crf 26" "intra_modes:--intra-modes" "subpel1:--subpel 1" "subpel2:--subpel 2" "deblock:--deblock" "deblock_gop_batch10:--deblock --gop-batch 10" "intra_modes_plane:--intra-modes 2" "intra4x4:--intra-modes 3" "all_tools:--crf 26 --intra-modes 3 --subpel 2 --deblock" "all_tools_gop_batch10:--crf 26 --intra-modes 3 --subpel 2 --deblock --gop-batch 10" "all_tools_r02_set_gop_batch10:--crf 26 --intra-modes 2 --subpel 2 --deblock --gop-batch 10" "preset_tools_no_filter:--crf 26 --intra-modes 3 --subpel 2"; do
  timeout -k 10 300 python bench.py --no-extras --cpu-frames 30 ${v#*:} > $O/bench_tool_${v%%:*}.json 2> $O/bench_tool_${v%%:*}.err; cut -c1-160 $O/bench_tool_${v%%:*}.json
done
timeout -k 10 300 python bench.py --workload 2160p --no-extras --cpu-frames 0 --subpel 2 > $O/bench_tool_subpel2_2160p.json 2> $O/bench_tool_subpel2_2160p.err
# sub-sample mode and the in-loop filter: kernel stats, every kernel alone on the chip
cd /tmp
VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tools -- python3 $R/bench.py --steps 2 --warmup 1 $Q --no-profile --subpel 2 --deblock --crf 26 --intra-modes 3 --gop-batch 10 > $O/stats_tools.log 2>&1
cd $R
cd /tmp
for wl in 1080p 2160p; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 2 $Q > $O/stats_$wl.log 2>&1
  echo "stats $wl done"
done
# SQ counters, every kernel alone on the chip (one stream)
for wl in 1080p 2160p; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/sq_$wl/${c%% *} -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 $Q --no-profile > $O/sq_$wl.log 2>&1
  done
  echo "sq $wl done"
done
cd $R
for wl in 1080p 2160p; do python3 tools/sq_summary.py $O/sq_$wl $wl profiles/sq_counters.json > $O/sq_counters_${wl}_isolated.txt 2> $O/sq_summary_$wl.err; done
cp profiles/sq_counters.json $O/
# the default line last in this half: it quotes the PMC traffic and the SQ counters collected above (same device sources)
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-240 $O/bench_default.json
fi
if [[ $PART == *b* ]]; then
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --mode slices --workload 2160p $Q > $O/bench_slices1.json 2> $O/bench_slices1.err; cut -c1-240 $O/bench_slices1.json
# two ranks on the one GPU (gloo rehearsal of the N > 1 run: bench.py starts its own ranks; the line carries the config-5 leg as slices_2160p)
timeout -k 10 500 python bench.py --gpus 2 --device 0 --backend gloo --steps 3 --warmup 1 $Q --slices-steps 2 > $O/bench_streams2.json 2> $O/bench_streams2.err
cut -c1-200 $O/bench_streams2.json
# the same under the driver's launcher
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --steps 3 --warmup 1 --gpus 2 --backend gloo --device 0 $Q --slices-steps 2 > $O/bench_streams2_torchrun.json 2> $O/bench_streams2_torchrun.err
grep -o '"equals_one_rank_bitstream": [a-z]*' $O/bench_streams2.json $O/bench_streams2_torchrun.json
# round 3: the search kernel's phase budget, instruction issue costs, ingest copy ceiling, the in-loop filter against waves per block
bash tools/me_phase_pmc.sh ${1:-final}/phase 1080p > /dev/null 2>&1; cp $O/phase/me_phase_1080p.txt $O/ 2>/dev/null
tools/valu_ubench > $O/valu_ubench.txt 2>&1
tools/h2d_ubench > $O/h2d_ubench.txt 2>&1
tools/ifetch_ubench > $O/ifetch_ubench.txt 2>&1
# same-box A/B of the round's kernels against the shapes they replace
bash tools/r03_defer2.sh > $O/ab_search_variants.txt 2>&1
bash tools/r03_dbk2.sh > $O/ab_deblock_kernels.txt 2>&1
bash tools/dbk_step_budget.sh ${1:-final}/dbkb > $O/deblock_step_budget.txt 2>&1
bash tools/r03_dbk_pmc.sh ${1:-final}/dbkpmc libvenc_hip.so > $O/deblock_sq_counters.txt 2>&1
fi
# keep the merged output small: the raw traces stay on the box
for d in $O/pmcf_* $O/pmcw_* $O/stats_* $O/sq_1080p $O/sq_2160p; do [ -d "$d" ] && find $d -name "*agent_info.csv" -delete; done
du -sh $O
