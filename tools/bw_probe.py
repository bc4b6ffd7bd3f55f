"""What the box writes / copies / reads on k_sobel's volume (6.46 GB of gradients out, 1.77 GB of image rows in):
the achievable side of profiles/r4_sobel_variants.md.  Run on the GPU box: python tools/bw_probe.py"""
import torch,time
x=torch.empty(6_460_000_000//2,dtype=torch.int16,device='cuda')
y=torch.empty(1_770_000_000,dtype=torch.uint8,device='cuda')
def t(f,n=5):
    f(); torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
print('zero_ 6.46GB ms', t(lambda: x.zero_()))
z=torch.empty_like(x)
print('copy 6.46GB->6.46GB ms', t(lambda: z.copy_(x)))
print('read-sum 1.77GB ms', t(lambda: y.sum()))
