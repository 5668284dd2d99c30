for k in 1 2; do
EZHIP_LIBRARY=$GRAFT_REPO_ROOT/devlibs/librmn_ez_hip_before.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print('before', round(b['ms_per_step'],4), round(b['extras']['single_field_launch_us'],2), round(b['pack']['us_per_field'],2), round(b['pack']['cfg5_pipeline_us_per_field'],2), round(b['extras']['cfg3_uvint']['us_per_pair'],1))"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print('after ', round(b['ms_per_step'],4), round(b['extras']['single_field_launch_us'],2), round(b['pack']['us_per_field'],2), round(b['pack']['cfg5_pipeline_us_per_field'],2), round(b['extras']['cfg3_uvint']['us_per_pair'],1))"
done
