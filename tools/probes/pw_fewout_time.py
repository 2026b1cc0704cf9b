import sys, os
sys.path.insert(0, os.getcwd())
import torch
from cenet_amd import kern
dev=torch.device('cuda:0')
B,Ci,Co,HW=32,64,4,12544
x=torch.randn(B,Ci,HW,device=dev).bfloat16(); W=(torch.randn(Co,Ci,device=dev)*0.1).bfloat16(); b=torch.randn(Co,device=dev)
y=torch.empty(B,Co,HW,device=dev,dtype=torch.bfloat16); g=torch.randn(B,Co,HW,device=dev).bfloat16(); dx=torch.empty_like(x)
dW=torch.zeros(Co,Ci,device=dev); db=torch.zeros(Co,device=dev)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print('fwd', t(lambda: kern.pw_fewout_fwd(x,W,b,y,B,Ci,Co,HW)))
print('dgrad', t(lambda: kern.pw_fewout_dgrad(g,W,dx,B,Ci,Co,HW)))
print('wgrad', t(lambda: kern.pw_fewout_wgrad(x,g,dW,db,B,Ci,Co,HW)))
