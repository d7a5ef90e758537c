import torch, time
n=446631
for (k,m) in [(240,480),(720,480),(240,240),(222,160),(74,74),(160,96)]:
    A=torch.randn(n,k,dtype=torch.float64,device='cuda'); C=torch.randn(k,m,dtype=torch.float64,device='cuda')
    for _ in range(2): (A@C)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): (A@C)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(f"n x {k} @ {k} x {m}: {dt*1e3:.2f} ms  {2*n*k*m/dt/1e12:.1f} TF/s")
