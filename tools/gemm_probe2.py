import torch, time
dev="cuda"
from hugectr_amd.dense import split_k_wgrad
for B in (16384, 65536):
    for (o,i) in [(512,13),(256,512),(128,256),(1024,480),(1024,1024),(512,1024),(256,512),(1,256)]:
        dy=torch.randn(B,o,device=dev,dtype=torch.bfloat16); x=torch.randn(B,i,device=dev,dtype=torch.bfloat16)
        for name,fn in (("split",lambda: split_k_wgrad(dy,x,16)),("plain",lambda: dy.t()@x)):
            fn(); torch.cuda.synchronize()
            ts=[]
            for _ in range(5):
                t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
            print(f"B={B} dW[{o},{i}] {name}: host+gpu ms per call min={min(ts):.3f} max={max(ts):.3f}")
