import sys, time, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/interactive-spectrogram-inpainting_amd')
import bench as Bn
dev=torch.device('cuda:0')
m,sd=Bn._build_model(dev)
x=torch.randn(64,2,128,512,generator=torch.Generator().manual_seed(100)).to(dev)
res={}
with torch.no_grad():
    for mode in ('f32','bf16x3_decoder','bf16x3','split_bf16','split_f16'):
        m.conv_precision=mode
        for _ in range(3): out=m(x)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): out=m(x)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
        res[mode]=(dt,[o.clone() for o in out]); print(mode, f"{dt*1e3:.3f} ms  {64/dt:.0f}/s")
ref=res['f32'][1]
for mode in ('bf16x3_decoder','bf16x3','split_bf16','split_f16'):
    o=res[mode][1]
    print(mode,"dec err",((o[0]-ref[0]).abs().max()/ref[0].abs().max()).item(),"id_t eq",(o[4]==ref[4]).float().mean().item(),"id_b eq",(o[5]==ref[5]).float().mean().item(), "n mismatches", int((o[4]!=ref[4]).sum()), int((o[5]!=ref[5]).sum()))
# against the CPU oracle on a small batch
from oracle import vqvae_oracle as O
cfg=O.Config(in_channel=2)
xs=x[:4].cpu()
with torch.no_grad(): oref=O.forward(xs, {k:v.cpu() for k,v in sd.items()}, cfg)
for mode in ('f32','bf16x3','split_bf16','split_f16'):
    m.conv_precision=mode
    with torch.no_grad(): o=m(x[:4])
    print(mode,"vs CPU oracle: id_t mismatches", int((o[4].cpu()!=oref[4]).sum()), "of", oref[4].numel(), " id_b", int((o[5].cpu()!=oref[5]).sum()), "of", oref[5].numel(), " dec err", float((o[0].cpu()-oref[0]).abs().max()/oref[0].abs().max()))
