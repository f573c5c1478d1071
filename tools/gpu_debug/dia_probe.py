"""Per-term difference of the DIAGNOSTICS_TS arrays, device against oracle (which term, which tracer, which step)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from tests import util
from tests.test_gpu_parity import _case_state

tag = sys.argv[1] if len(sys.argv) > 1 else "upwelling_small"
kw = dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")) if tag == "upwelling_small" else {}
cs, g = _case_state(tag, kw)
O = util.make_oracle(cs, g)
H = util.make_hip(cs, g)
O.set_dia_window(3, 1)
H.dia_config(3, 1)
O.start()
H.start()
NT, N = 2, cs["N"]
for step in range(1, 4):
    O.main3d_step()
    H.main3d(1)
    for n in ("DiaTwrk", "DiaTrc"):
        a, b = H.download(n), O.field(n)
        ndt = a.size // (NT * N * (a.size // (NT * N * (a.size // (NT * N)))) ) if False else None
        plane = a.size // (b.size // 1) if False else None
        nplanes = None
        a = a.reshape(-1, NT, N, a.size // (NT * N) // (a.size // (NT * N) // 1) if False else 1) if False else a
        tot = a.size
        # (idiag, itrc, k, j*i)
        per = tot // (NT * N)
        for ndt_try in (6, 9, 10):
            if per % ndt_try == 0 and abs(per // ndt_try - (cs["Lm"] + 6) * (cs["Mm"] + 6)) < 400:
                ndt = ndt_try
        A, B = a.reshape(ndt, NT, N, -1), b.reshape(ndt, NT, N, -1)
        for d in range(ndt):
            for it in range(NT):
                e = np.abs(A[d, it] - B[d, it]).max()
                s = np.abs(B[d, it]).max()
                if e > 1e-10 * max(s, 1e-30):
                    k = np.unravel_index(np.argmax(np.abs(A[d, it] - B[d, it])), A[d, it].shape)
                    print(f"step {step} {n} term {d} tracer {it}: err {e:.3e} of {s:.3e} at k={k[0]} p={k[1]}", flush=True)
    for n in ("t", "u"):
        print(step, n, util.relrms(H.download(n), O.field(n)))
H.close()
