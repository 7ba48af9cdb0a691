import torch, time
for nbytes in (1_345_200_000, 13_452_000_000):
    x = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    for _ in range(2): x.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): x.zero_()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"fill {nbytes/1e9:.2f} GB: {ms:.3f} ms = {nbytes/ms/1e9:.2f} TB/s", flush=True)
    y = torch.empty_like(x) if nbytes < 5e9 else None
    if y is not None:
        e0.record()
        for _ in range(5): y.copy_(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"copy {nbytes/1e9:.2f} GB: {ms:.3f} ms = {2*nbytes/ms/1e9:.2f} TB/s (read+write)", flush=True)
    del x, y
