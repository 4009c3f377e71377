import sys
sys.path[:0]=['/root/repo','/root/repo/gs-lora_amd']
import torch
from gslora_hip import ops
for dt in (torch.bfloat16, torch.float16):
    for (M,N,K) in ((33490,512,512),(300,256,64)):
        for val in (1.0, 200.0):
            A=torch.full((M,K),val,device='cuda',dtype=dt); W=torch.full((N,K),val,device='cuda',dtype=dt)
            A[5,0]=float('nan'); A[6,1]=float('inf')
            out=torch.empty(M,N,device='cuda',dtype=dt)
            ops.gemm_nt(A,W,out)
            print(dt,M,N,K,val,"row5",out[5,:3].tolist(),"row6",out[6,:3].tolist(),"row7",out[7,:3].tolist())
