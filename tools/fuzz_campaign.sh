# the round's fuzz campaign (one gpurun call): new seeds every round -- seed base as $1
S=${1:-600}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
run() { echo "== $*"; python "$@" 2>&1 | tail -2; }
run tools/fuzz_vs_ref2.py 300 $((S+1))
run tools/fuzz_vs_ref2.py 300 $((S+2))
FUZZ_HEMI=1 python tools/fuzz_vs_ref2.py 200 $((S+3)) 2>&1 | tail -2
run tools/fuzz_vs_ref.py 300 $((S+4))
run tools/fuzz_vs_ref3.py 300 $((S+5))
run tools/fuzz_vs_ref4.py 80 $((S+6))
run tools/fuzz_armn32.py 200 $((S+7))
run tools/fuzz_armn_wide.py 150 $((S+8))
run tools/fuzz_armn.py 100 $((S+9))
run tools/fuzz_average.py 80 $((S+10))
run tools/fuzz_interpv.py 80 $((S+11))
run tools/fuzz_shapes.py $((S+12)) 60
