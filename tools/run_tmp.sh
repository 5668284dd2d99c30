mkdir -p gpurun_out/r5m
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5m/tests.txt
python -m pytest tests/test_gpu_packers.py -q -m gpu -k "cfg5_record_against or fused_cfg5" -s 2>&1 | grep -i "cfg5 end\|passed\|failed" >> gpurun_out/r5m/tests.txt
python bench.py > gpurun_out/r5m/bench.json 2> gpurun_out/r5m/bench.err
cat gpurun_out/r5m/tests.txt; tail -3 gpurun_out/r5m/bench.err; python - <<'PY'
import json
b=json.loads(open('gpurun_out/r5m/bench.json').readline())
print('ms_per_step', b['ms_per_step'], 'frac', b['roofline']['frac'], 'checked', b['checked'])
e=b['extras']; print('single', e['single_field_launch_us'], 'cfg3 uv', e['cfg3_uvint']['us_per_pair'], 'first', e.get('cfg3_first_call_ms'), e.get('cfg3_second_call_ms'), 'sint', e['cfg3_sint']['us_per_field'])
print('pack', b['pack']['us_per_field'], 'pipe', b['pack']['cfg5_pipeline_us_per_field'], b['pack']['roofline_cfg5'])
print(b['roofline_single_field'])
PY
