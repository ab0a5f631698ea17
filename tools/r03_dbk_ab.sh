#!/bin/bash
# A/B of library variants on the in-loop filter: bash tools/r03_dbk_ab.sh <tag> var1.so var2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/video-encoder_amd/csrc
O=$R/gpurun_out/$1; shift; mkdir -p $O
cd $R
cp $C/libvenc_hip.so $O/orig.so
for v in "$@"; do
  cp $C/$v $C/libvenc_hip.so
  timeout -k 10 300 python -m pytest tests/test_gpu_deblock.py -x -q > $O/pytest_${v%.so}.log 2>&1 || { echo "$v: deblock tests FAILED"; tail -5 $O/pytest_${v%.so}.log; continue; }
  for t in "--deblock" "--deblock --gop-batch 30" "--crf 26 --intra-modes 2 --subpel 2 --deblock --gop-batch 30"; do
    timeout -k 10 250 python bench.py --no-extras --cpu-frames 30 --steps 6 $t > $O/dbk.json 2> $O/dbk.err || { echo "$v $t FAILED"; tail -3 $O/dbk.err; continue; }
    python3 - $O/dbk.json "$v $t" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ki = d.get("kernel_ms_per_step_isolated", {})
print("%-74s %9.1f Mpx/s %7.3f ms/step bit_exact %s | iso k_deblock=%.2f k_intra=%.2f" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("bit_exact"), ki.get("k_deblock", 0), ki.get("k_intra", 0)))
PY
  done
done
cp $O/orig.so $C/libvenc_hip.so
