"""how long the 32-field launch takes as a function of how long the device has been busy: blocks of 60 launches, back to back, from a cold process (development)"""
import os, sys, time
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
ez.ezsint_batch_dev(d_out, d_in, F); torch.cuda.synchronize()
t_start = time.perf_counter()
done = 0
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(60): ez.ezsint_batch_dev(d_out, d_in, F)
    e1.record(stream); torch.cuda.synchronize()
    done += 60
    if blk < 6 or blk % 6 == 5:
        print(f"launches {done - 60:5d} .. {done:5d}  busy for {time.perf_counter() - t_start:5.2f} s: {e0.elapsed_time(e1) * 1e3 / 60 / F:6.2f} us per field", flush=True)
