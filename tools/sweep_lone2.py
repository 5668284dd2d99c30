"""A lone cfg2 field per launch (c_ezsint_dev, bicubic, polar correction on) under development switches, INTERLEAVED on one device: us per launch.
   python tools/sweep_lone2.py "VAR=val" ...  ("" = defaults)"""
import os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import torch
from librmn_amd import ezscint as ez
import ezcases as ec
NI_S, NJ_S, NI_D, NJ_D = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(NI_S, NJ_S, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(NI_D, NJ_D, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
d_in = torch.from_numpy(ec.synth_field(NI_S, NJ_S, seed=1)).cuda()
d_out = torch.empty(NI_D * NJ_D, dtype=torch.float32, device="cuda")
def t(reps, warm):
    for _ in range(warm): ez.ezsint_dev(d_out, d_in)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): ez.ezsint_dev(d_out, d_in)
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
variants = sys.argv[1:] or [""]
t(300, 300)
res = {v: [] for v in variants}; touched = set()
for rnd in range(7):
    for v in variants:
        for k in touched: os.environ.pop(k, None)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("="); os.environ[k] = val; touched.add(k)
        res[v].append(t(100, 10))
for v in variants:
    r = sorted(res[v]); print(f"[{v or 'defaults'}] median {r[3]:.2f} min {r[0]:.2f} max {r[-1]:.2f} us per lone field", flush=True)
