#!/bin/bash
# the shim's new defaults: the tests that open the encoder as the reference does, then the whole bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_c7}; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "reference_open or long_key_frame or c_driver or av_encode or host_frames or rotating or x264_api or io" > $O/test.log 2>&1; rc=$?
tail -6 $O/test.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?
tail -3 $O/bench.err
python3 - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "2160p", d.get("2160p", {}).get("value"), "bit_exact", d.get("bit_exact"))
print("roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "isolated_frac", "traffic")})
print("reference_defaults", json.dumps(d.get("reference_defaults"))[:900])
print("config", json.dumps(d["config"])[:1500])
PY
exit $rc
