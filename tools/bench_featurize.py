#!/usr/bin/env python3
"""HBM bandwidth of the stand-alone featurise kernel (tg_featurize_dev)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tamago_amd import lib as tl
lib = tl.load()
for size in (9, 19):
    p = size * size
    b = (1 << 21) if size == 9 else (1 << 19)
    cells = torch.randint(0, 3, (b, p), dtype=torch.uint8, device="cuda")
    tm = torch.randint(1, 3, (b,), dtype=torch.int8, device="cuda")
    prev = torch.randint(0, (size + 2) ** 2, (b,), dtype=torch.int32, device="cuda")
    mv = torch.randint(1, 50, (b,), dtype=torch.int32, device="cuda")
    out = torch.empty((b, 6, size, size), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def run():
        tl.check(lib.tg_featurize_dev(size, cells.data_ptr(), tm.data_ptr(), prev.data_ptr(), mv.data_ptr(), b, out.data_ptr(), st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    bytes_ = b * (p + 9 + 6 * p * 4)
    print(f"featurize S={size} B={b}: {ms*1e3:.1f} us, {bytes_/ms/1e6:.1f} GB/s algorithmic "
          f"({bytes_/ms/1e6/8000*100:.1f}% of 8 TB/s), {b/ms*1e3/1e6:.1f} M positions/s")
