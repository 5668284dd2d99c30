import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from librmn_amd import ezscint as ez
import ezcases as ec
NI_S, NJ_S, NI_D, NJ_D = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(NI_S, NJ_S, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(NI_D, NJ_D, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
d_in = torch.from_numpy(ec.synth_field(NI_S, NJ_S, seed=1)).cuda()
d_out = torch.empty(NI_D * NJ_D, dtype=torch.float32, device="cuda")
for deg in ("cubic", "nearest"):
    for polar in ("yes", "no"):
        ez.ezsetopt("interp_degree", deg); ez.ezsetopt("polar_correction", polar)
        for _ in range(60): ez.ezsint_dev(d_out, d_in)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(200): ez.ezsint_dev(d_out, d_in)
        e1.record(stream); torch.cuda.synchronize()
        print(f"single-field c_ezsint_dev {deg} polar={polar}: {e0.elapsed_time(e1)*1e3/200:.1f} us")
