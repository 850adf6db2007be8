#!/bin/bash
# round 4: the default bench line as the driver runs it, with a summary of the new legs (run through gpurun from the repository root)
O=gpurun_out/${1:-r04e}
mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; echo rc=$?
tail -5 $O/bench.log; grep "^\[bench\]" $O/bench.log | cut -c1-300 | tail -40
python3 - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_extended"], d["wall_s_total"])
print(json.dumps(d["oracle_check"])[:600])
print(json.dumps(d["cpu_baseline"])[:1500])
e = d["extras"]
print(json.dumps(e["c3_pairs"].get("end_to_end"))[:1500])
print(json.dumps(e["c3_pairs"].get("bwt_stage_resident"))[:800])
print(json.dumps(e.get("c5_pairs"))[:2500])
print(json.dumps(e.get("repeat_models"))[:1500])
PY
