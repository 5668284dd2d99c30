cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5f/gputests.txt
python bench.py > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err
cat gpurun_out/r5f/gputests.txt; tail -3 gpurun_out/r5f/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"])
e = d["extras"]
for k in ("first_call_setup_ms", "first_call_setup_note", "cfg3_first_call_ms", "cfg3_second_call_ms"):
    print(k, e.get(k))
for k in ("cfg3_uvint", "cfg3_sint", "armn_uncompress", "armn_uncompress_minimum", "armn_uncompress_minimum_ragged", "armn_uncompress32", "armn_uncompress32_whole_tile_rows"):
    print(k, {a: b for a, b in e.get(k, {}).items() if not isinstance(b, (str, dict))})
print("pack", {a: b for a, b in d["pack"].items() if not isinstance(b, (str, dict))})
PY
