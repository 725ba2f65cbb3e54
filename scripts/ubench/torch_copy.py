import torch, time
dev=torch.device('cuda',0)
for nbytes in [1<<30, 3<<30]:
    x=torch.empty(nbytes//4, dtype=torch.float32, device=dev).normal_()
    y=torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"torch copy {nbytes/1e9:.2f} GB: {ms:.3f} ms  {2*nbytes/ms/1e6:.0f} GB/s")
    z=torch.empty(nbytes//8, dtype=torch.float32, device=dev)
    e0.record()
    for _ in range(10): torch.mul(x[:nbytes//8], 2.0, out=z)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"torch mul {nbytes/8/1e9:.2f} G: {ms:.3f} ms  {2*nbytes/2/ms/1e6:.0f} GB/s")
