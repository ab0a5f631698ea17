#!/bin/bash
# Measurement batch for profiles/: run on the GPU box from the repo root:  bash tools/final_batch.sh <tag>
# PMC passes are separate rocprofv3 runs (counters never together with --stats); the program itself follows "--".
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/test.log 2>&1; tail -2 $O/test.log
cd /tmp; export TMPDIR=/tmp
for wl in 1080p 2160p; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$wl -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --cpu-frames 0 --no-profile > $O/pmcf_$wl.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$wl -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --cpu-frames 0 --no-profile > $O/pmcw_$wl.log 2>&1
done
cd $R
for wl in 1080p 2160p; do python3 tools/pmc_summary.py $O/pmcf_$wl $O/pmcw_$wl $wl > /dev/null; done
cp profiles/pmc_traffic.json $O/
timeout -k 10 300 python bench.py > $O/bench_1080p.json 2> $O/bench_1080p.err; cut -c1-240 $O/bench_1080p.json
timeout -k 10 300 python bench.py --workload 2160p --cpu-frames 0 > $O/bench_2160p.json 2> $O/bench_2160p.err; cut -c1-240 $O/bench_2160p.json
timeout -k 10 300 python bench.py --serial-steps --cpu-frames 0 > $O/bench_1080p_serial.json 2> $O/bench_1080p_serial.err
timeout -k 10 300 python bench.py --serial-steps --workload 2160p --cpu-frames 0 > $O/bench_2160p_serial.json 2> $O/bench_2160p_serial.err
cd /tmp
for wl in 1080p 2160p; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --cpu-frames 0 > $O/stats_$wl.log 2>&1
done
cd $R
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --mode slices --workload 2160p --cpu-frames 0 > $O/bench_slices1.json 2> $O/bench_slices1.err; cut -c1-240 $O/bench_slices1.json
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --steps 3 --warmup 1 --gpus 2 --mode slices --workload 2160p --backend gloo --device 0 --cpu-frames 0 > $O/bench_slices2.json 2> $O/bench_slices2.err
grep -o '"bitstream_sha256": "[0-9a-f]*"' $O/bench_slices1.json $O/bench_slices2.json
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --steps 3 --warmup 1 --gpus 2 --backend gloo --device 0 --cpu-frames 0 > $O/bench_streams2.json 2> $O/bench_streams2.err
cut -c1-200 $O/bench_streams2.json
