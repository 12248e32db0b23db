import torch, time
B,n2,n=2048,768,384
X=torch.randn(B,n2,dtype=torch.float64,device='cuda'); K=torch.randn(n,n2,dtype=torch.float64,device='cuda')
Kt=K.t().contiguous()
U=torch.empty(B,n,dtype=torch.float64,device='cuda')
for name,fn in (("X@K.T",lambda: torch.matmul(X,K.t(),out=U)),("X@Kt(contig)",lambda: torch.matmul(X,Kt,out=U))):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/200
    print(name,"%.1f us  %.1f TF"%(us,2*B*n2*n/us*1e-6))
