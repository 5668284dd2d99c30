"""Ad-hoc timing probe for k_sep variants on the cfg2 shape (development tool, not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from librmn_amd import ezscint as ez
import ezcases as ec

NI_S, NJ_S, NI_D, NJ_D = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(NI_S, NJ_S, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(NI_D, NJ_D, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
F = 16
base = torch.from_numpy(ec.synth_field(NI_S, NJ_S, seed=1)).cuda()
d_in = torch.stack([base + 0.01 * f for f in range(F)]).contiguous()
d_out = torch.empty((F, NI_D * NJ_D), dtype=torch.float32, device="cuda")


def timeit(fn, reps=40):
    for _ in range(10):          # past the clock ramp of the first launches
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / F


DEGS = sys.argv[1].split(",") if len(sys.argv) > 1 else ("cubic", "linear", "nearest")
for deg in DEGS:
    for polar in (tuple(os.environ.get("PROBE_POLAR", "yes").split(",")) if len(sys.argv) > 1 else ("yes", "no")):
        ez.ezsetopt("interp_degree", deg); ez.ezsetopt("polar_correction", polar)
        ez.prepare_set()
        us = timeit(lambda: ez.ezsint_batch_dev(d_out, d_in, F))
        print(f"k_sep {deg:8s} polar={polar:3s}: {us:8.2f} us/field  {142.4288e6 / us / 1e3:8.1f} GB/s algorithmic")
if len(sys.argv) > 1:
    sys.exit(0)
# reference points: device copies of the same byte volumes
src = d_in.view(-1); dst = torch.empty_like(src)
us = timeit(lambda: dst.copy_(src)) ; print(f"copy 38.7MB->38.7MB x{F}: {us:8.2f} us/field-equivalent")
big = d_out.view(-1); big2 = torch.empty_like(big)
us = timeit(lambda: big2.copy_(big)); print(f"copy 103.7MB->103.7MB x{F}: {us:8.2f} us/field  ({2*103.7088e6/us/1e3:.0f} GB/s)")
us = timeit(lambda: big.fill_(1.0)); print(f"fill 103.7MB x{F}: {us:8.2f} us/field ({103.7088e6/us/1e3:.0f} GB/s)")
