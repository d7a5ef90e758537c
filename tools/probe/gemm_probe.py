import torch, time
n=446631
for (k,m) in [(240,480),(720,480),(240,240),(222,160),(74,74),(160,96)]:
    A=torch.randn(n,k,dtype=torch.float64,device='cuda'); C=torch.randn(k,m,dtype=torch.float64,device='cuda')
    for _ in range(2): (A@C)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): (A@C)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(f"n x {k} @ {k} x {m}: {dt*1e3:.2f} ms  {2*n*k*m/dt/1e12:.1f} TF/s")
print("Gram shapes  X^T Y")
for (a,b) in [(240,240),(480,240),(80,80),(160,96)]:
    X=torch.randn(n,a,dtype=torch.float64,device='cuda'); Y=torch.randn(n,b,dtype=torch.float64,device='cuda')
    for _ in range(2): (X.T@Y)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): (X.T@Y)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(f"(n x {a})^T (n x {b}): {dt*1e3:.2f} ms  {2*n*a*b/dt/1e12:.1f} TF/s  {8*n*(a+b)/dt/1e12:.2f} TB/s")
