"""How large a single hipMalloc does this box allow?  (the 2^27-point window table is 103 GB)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import za_amd
ctx = za_amd.Context(0)
print("mem_get_info", [x / 1e9 for x in torch.cuda.mem_get_info()])
for gb in (40, 64, 80, 96, 103, 112, 128, 160, 200):
    try:
        d = ctx.alloc(int(gb * 1e9))
        print(gb, "GB ok")
        ctx.free(d)
    except Exception as e:
        print(gb, "GB failed:", e)
