#!/usr/bin/env python3
"""hipMemsetAsync captured by torch.cuda.graph: does the memset NODE run on every replay?  Captured sequence on one tensor:
x.fill_(7) [kernel] -> hipMemsetAsync(x, 0) [runtime memset node] -> x.add_(1) [kernel]; x must be 1 after every replay (8 = the node did
nothing).  Variant B leaves out the leading fill: the node is then the first writer of x in the graph, as the gradient / counter zeroing of the
captured training step is (x accumulates across replays if the node does not run: 1, 2, 3, ...)."""
import ctypes, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device("cuda:0")
for words in (2240, 3_537_920, 4_300_800):
    for lead_fill in (True, False):
        x = torch.zeros(words, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            x.add_(1)                         # warm-up of the torch kernels outside the capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        x.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s = torch.cuda.current_stream(dev).cuda_stream
            if lead_fill:
                x.fill_(7)
            rc = hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, ctypes.c_size_t(words * 4), ctypes.c_void_p(s))
            assert rc == 0, rc
            x.add_(1)
        vals = []
        for rep in range(4):
            g.replay()
            torch.cuda.synchronize()
            vals.append((int(x.min()), int(x.max())))
        ok = all(v == (1, 1) for v in vals)
        print(f"{words * 4:>10d} bytes, {'fill -> memset node -> add' if lead_fill else 'memset node -> add':28s}: (min, max) after replays {vals}  {'ok' if ok else 'MEMSET NODE NOT EXECUTED ON EVERY REPLAY'}")
