import torch, time
x = torch.empty(int(6.3e9)//4, dtype=torch.float32, device='cuda')
for _ in range(3): x.fill_(1.0)
torch.cuda.synchronize()
ts=[]
for _ in range(10):
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); x.fill_(2.0); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
print("fill_ GB/s", x.numel()*4/ (min(ts)*1e-3)/1e9, "median", x.numel()*4/(sorted(ts)[5]*1e-3)/1e9)
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
ts=[]
for _ in range(10):
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); y.copy_(x); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
print("copy_ GB/s (r+w)", 2*x.numel()*4/(min(ts)*1e-3)/1e9)
