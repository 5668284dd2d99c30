"""Lone-field c_ezsint_dev (cfg2, polar on) after the clocks have ramped: us per launch for EZHIP_SINGLE_RB = row-blocks per thread block."""
import os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import torch
from librmn_amd import ezscint as ez
import ezcases as ec
NI_S, NJ_S, NI_D, NJ_D = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(NI_S, NJ_S, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(NI_D, NJ_D, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
F = 32
base = torch.from_numpy(ec.synth_field(NI_S, NJ_S, seed=1)).cuda()
d_in = torch.stack([base + 0.01 * f for f in range(F)]).contiguous()
d_out = torch.empty((F, NI_D * NJ_D), dtype=torch.float32, device="cuda")
def t(fn, reps, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("batch of 32, per field:", round(t(lambda: ez.ezsint_batch_dev(d_out, d_in, F), 60, 40) / F, 2))
for rb in sys.argv[1:] or ["0", "2", "3", "4", "5", "6", "7", "9"]:
    if rb == "0": os.environ.pop("EZHIP_SINGLE_RB", None)
    else: os.environ["EZHIP_SINGLE_RB"] = rb
    print(f"EZHIP_SINGLE_RB={rb}: {t(lambda: ez.ezsint_dev(d_out[0], d_in[0]), 100, 10):.2f} us per lone field", flush=True)
