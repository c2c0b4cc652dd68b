import torch, time
dev=torch.device("cuda:0")
for (M,N,K) in [(65536,2048,512),(29785,1200,300),(4096,4096,4096)]:
    A=torch.randn(M,K,device=dev); B=torch.randn(N,K,device=dev)
    for name,fn in [("matmul A@B.T", lambda: A@B.T), ("F.linear", lambda: torch.nn.functional.linear(A,B))]:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print(f"{name:14s} {M}x{N}x{K}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TF", flush=True)
print(torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision())
