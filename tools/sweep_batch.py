"""The 32-field cfg2 launch (c_ezsint_batch_dev, bicubic, polar correction on) under development switches, INTERLEAVED on one device (a device drifts by 3 % within
minutes): us per field.  python tools/sweep_batch.py "VAR=val" "VAR=val,VAR2=val" ...  ("" = defaults)"""
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
def t(reps, warm):
    for _ in range(warm): ez.ezsint_batch_dev(d_out, d_in, F)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): ez.ezsint_batch_dev(d_out, d_in, F)
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / F
variants = sys.argv[1:] or [""]
t(40, 40)
res = {v: [] for v in variants}
touched = set()
for rnd in range(5):
    for v in variants:
        for k in touched: os.environ.pop(k, None)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("="); os.environ[k] = val; touched.add(k)
        res[v].append(t(30, 5))
for v in variants:
    r = sorted(res[v]); print(f"[{v or 'defaults'}] median {r[2]:.2f} min {r[0]:.2f} max {r[-1]:.2f} us per field", flush=True)
