cd $GRAFT_REPO_ROOT
for g in 30 40 50 64; do
  timeout -k 10 250 python bench.py --no-extras --cpu-frames 0 --crf 26 --intra-modes 3 --subpel 2 --deblock --gop-batch $g --steps 16 > gpurun_out/r04_items_$g.json 2> gpurun_out/r04_items_$g.err || { echo FAIL $g; tail -3 gpurun_out/r04_items_$g.err; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_items_$g.json').read().strip().splitlines()[-1]); print('all tools, items $g:', d['value'], d['ms_per_step'])"
done
for g in 30 64; do
  timeout -k 10 250 python bench.py --no-extras --cpu-frames 0 --deblock --gop-batch $g --steps 16 > gpurun_out/r04_items_d$g.json 2> gpurun_out/r04_items_d$g.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_items_d$g.json').read().strip().splitlines()[-1]); print('deblock only, items $g:', d['value'], d['ms_per_step'])"
done
