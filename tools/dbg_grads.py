import sys, os, numpy as np, torch
REPO="/root/repo"; sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,"diffusion-deconvolution-dia-msms-data_amd")); sys.path.insert(0, os.path.join(REPO,"tests"))
from dquartic.model.unet1d import UNet1d
z=np.load(os.path.join(REPO,"tests/golden/unet_default_rt16.npz"))
sd={k[2:]:torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
net=UNet1d(dim=4,channels=1,dim_mults=(1,2,2,3,3,4,4),conditional=True,init_cond_channels=1,attn_cond_channels=1,downsample_dim=64,simple=True)
net.load_state_dict(sd); net=net.cuda()
T=torch.from_numpy
x=T(z["x"]).cuda().requires_grad_()
y=net(x,T(z["t"]).cuda(),T(z["init_cond"]).cuda(),T(z["attn_cond"]).cuda())
(y*T(z["gout"]).cuda()).sum().backward(); torch.cuda.synchronize()
named=dict(net.named_parameters())
errs=[]
for k in z.files:
    if k.startswith("rope/grad/"):
        n=k[len("rope/grad/"):]; v=T(z[k]); e=float((named[n].grad.cpu()-v).abs().max()/max(float(v.abs().max()),1e-6))
        errs.append((e,n))
errs.sort(reverse=True)
for e,n in errs[:25]: print(f"{e:.3e} {n}")
print("dx", float((x.grad.cpu()-T(z["rope/dx"])).abs().max()/T(z["rope/dx"]).abs().max()))
