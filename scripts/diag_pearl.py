import sys, os, hashlib, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breaching_amd import _lib
lib = _lib.load()
print("lib", _lib.library_path(), hashlib.md5(open(_lib.library_path(), "rb").read()).hexdigest(), "abi", lib.bh_abi_version())
dev = torch.device("cuda:0")
rows = torch.zeros(4 * 4, dtype=torch.float64, device=dev)
rows[0], rows[1], rows[2] = 0.002, 34.0, 50.0
rows[4], rows[5], rows[6] = 0.001, 2.0, 1.0
for kind in (7, 4, 0):
    for fd in (0.0, 1e-3, 0.5, 2.0):
        stats = torch.full((12,), -7.0, device=dev)
        rc = lib.bh_gm_finalize(kind, _lib.ptr(rows), 2, 0.7, 0.0, 1e-7, fd, _lib.ptr(stats), None, _lib.current_stream_handle(dev))
        torch.cuda.synchronize()
        print(kind, fd, rc, [round(v, 6) for v in stats.cpu().tolist()])
