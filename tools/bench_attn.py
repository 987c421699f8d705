import torch, torch.nn.functional as F, time
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for (B,H,S,hd,causal,dt) in [(8,32,900,96,True,torch.bfloat16),(8,16,577,64,False,torch.float16),(8,16,577,64,False,torch.bfloat16)]:
    qkv = torch.randn(B,S,3*H,hd,device="cuda").to(dt)
    q,k,v = qkv[:,:,:H],qkv[:,:,H:2*H],qkv[:,:,2*H:]
    f1 = lambda: F.scaled_dot_product_attention(q.transpose(1,2),k.transpose(1,2),v.transpose(1,2),is_causal=causal).transpose(1,2).contiguous()
    qc,kc,vc = (t.transpose(1,2).contiguous() for t in (q,k,v))
    f2 = lambda: F.scaled_dot_product_attention(qc,kc,vc,is_causal=causal)
    fl = 4*B*H*S*S*hd/(2 if causal else 1)
    t1,t2 = timeit(f1),timeit(f2)
    print(f"B{B} H{H} S{S} hd{hd} causal={causal} {dt}: strided+contig-out {t1*1e3:.1f} us ({fl/t1/1e9:.0f} TF/s)  contiguous {t2*1e3:.1f} us ({fl/t2/1e9:.0f} TF/s)")

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynam3d_amd.hip_dense import HipDense
hd_ = HipDense()
for vtr in (False, True, False, True):
  HipDense.V_TR = vtr
  print("== V_TR", vtr)
  for (B,H,S,hd,causal,dt) in [(8,32,900,96,True,torch.bfloat16),(8,16,577,64,False,torch.float16),(8,16,577,64,False,torch.bfloat16),(2,4,200,96,True,torch.bfloat16),(8,32,900,96,False,torch.bfloat16)]:
      qkv = torch.randn(B,S,3*H,hd,device="cuda").to(dt)
      q,k,v = qkv[:,:,:H],qkv[:,:,H:2*H],qkv[:,:,2*H:]
      ref = F.scaled_dot_product_attention(q.transpose(1,2).float(),k.transpose(1,2).float(),v.transpose(1,2).float(),is_causal=causal).transpose(1,2)
      got = hd_.attention_qkv(qkv, H, causal).float()
      err = float((got-ref).norm()/ref.norm())
      fl = 4*B*H*S*S*hd/(2 if causal else 1)
      t = timeit(lambda: hd_.attention_qkv(qkv, H, causal))
      print(f"own flash B{B} H{H} S{S} hd{hd} causal={causal} {dt}: {t*1e3:.1f} us ({fl/t/1e9:.0f} TF/s) relerr {err:.2e}")
