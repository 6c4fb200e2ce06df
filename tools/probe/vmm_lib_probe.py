import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
def trial(name, rows_list, d=768, use_host=False):
    import numpy as np
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 0))
    buf = torch.zeros((max(rows_list), d), device="cuda")
    ok = []
    for r in rows_list:
        try:
            if use_host:
                idx.Add(None, np.zeros((r, d), np.float32))
            else:
                idx.add_device(r, buf.data_ptr())
            ok.append("ok")
        except Exception as e:
            ok.append("FAIL:" + str(e)[-40:])
            break
    import numpy as np
    lab, dist = idx.SearchBatch(np.zeros((1, d), np.float32), 5)
    print(name, rows_list, ok, "ntotal", idx.ntotal, "search", lab[0].tolist(), flush=True)
    idx.Close()
trial("dev 100k x2", [100_000, 100_000])
trial("dev 1M x3", [1_000_000] * 3)
trial("dev 1M x3 again", [1_000_000] * 3)
trial("dev small growth", [10, 10, 1000, 100000, 500000])
trial("host 100k x3", [100_000] * 3, use_host=True)
trial("dev 100k x2 d=1024 (1 MiB rows... 4096 B rows)", [100_000, 100_000], d=1024)
trial("dev 65536 rows x3 d=1024 (256 MiB chunks)", [65536] * 3, d=1024)
