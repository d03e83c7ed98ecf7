import torch
SHAPES = {"7b":[("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)],
          "70b-tp8":[("wqkv", 1280, 8192), ("wo", 8192, 1024), ("w13", 7168, 8192), ("w2", 8192, 3584)]}
for model, Ms in (("7b",(16,64,128,256,512)),("70b-tp8",(64,256))):
  for M in Ms:
    tt=tf=0
    for name,N,K in SHAPES[model]:
        a=torch.randn(M,K,device="cuda",dtype=torch.float16); b=torch.randn(N,K,device="cuda",dtype=torch.float16)
        for _ in range(3): c=a@b.t()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(20): c=a@b.t()
        e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/20; tt+=t; tf+=2.0*M*N*K
    print(f"hipBLASLt fp16 {model} M={M}: layer {tt*1e3:.1f} us ({tf/tt/1e9:.0f} TFLOP/s)")
