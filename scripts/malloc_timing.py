import time, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
for gb in (1, 4, 16, 34):
    torch.cuda.synchronize(); t0 = time.time()
    x = torch.empty(int(gb * 2**30), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t1 = time.time()
    del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t2 = time.time()
    print("hipMalloc %2d GB: %.1f ms, free %.1f ms" % (gb, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
