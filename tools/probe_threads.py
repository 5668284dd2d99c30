# two host threads, each with its own stream, interpolate different fields on the SAME grid set concurrently (first use included)
import sys, threading
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, ctypes
from librmn_amd import ezscint as ez
import oraclelib as ol, ezcases as ec
ni, nj, no, mo = 360, 181, 777, 391
gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 46, 46, 0, 0)
O = ol.oracle(); gs = O.orc_defset(ol.grid_define(no, mo, "L", (46, 46, 0, 0)), ol.grid_define(ni, nj, "G"))
res = {}
def work(tid):
    st = torch.cuda.Stream()
    ez.ezdefset(gdout, gdin)                     # thread-local current set
    ez.use_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        outs = []
        for k in range(6):
            f = ec.synth_field(ni, nj, seed=500 + 10 * tid + k)
            d_in = torch.from_numpy(f).cuda(non_blocking=False)
            d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda")
            assert ez.ezsint_dev(d_out, d_in) == 0
            outs.append((f, d_out))
        st.synchronize()
    res[tid] = outs
ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
bad = 0
opts = ol.default_opts()
for tid in res:
    for f, d_out in res[tid]:
        want = np.zeros(no * mo, np.float32)
        O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(f))
        got = d_out.cpu().numpy()
        err = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
        bad += int(err.max() > 1e-5)
print("multi-thread fields checked:", sum(len(v) for v in res.values()), "bad:", bad)
