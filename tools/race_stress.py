"""Schedule stress: the same fused train step (same inputs, t, noise; gradients cleared) repeated N times must give bit-identical loss and
gradients every time -- the main / side queue schedule of dq_train_step shares scratch buffers between the queues, and a missing ordering
shows up as a run-to-run difference.  usage: tools/race_stress.py [N]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic.model.model import DDIMDiffusionModel
from dquartic.model.unet1d import UNet1d

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(5)
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
with torch.no_grad():
    for _, p in net.trainable_named():
        p.add_(torch.randn_like(p) * 0.05)
dm = DDIMDiffusionModel(model_class=net, device="cuda")
bad_total = 0
for B, RT in ((1, 400), (3, 70), (8, 400), (32, 400)):
    g = torch.Generator().manual_seed(B)
    x0 = torch.rand(B, RT, 64, generator=g).cuda(); c2 = torch.rand(B, RT, 64, generator=g).cuda(); c1 = torch.rand(B, RT, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda(); noise = torch.randn(B, RT, 64, generator=g).cuda()
    ref_loss = ref = None
    bad = 0
    for i in range(N):
        loss = dm.train_step_fused(x0, c2, c1, t=t, noise=noise, zero_grads=True)
        gr = net.flat_grads()
        if ref is None:
            ref, ref_loss = gr.clone(), loss.clone()
        elif not (torch.equal(gr, ref) and torch.equal(loss, ref_loss)):
            bad += 1
    torch.cuda.synchronize()
    print(f"B={B} RT={RT}: {N} identical steps, {bad} differ from the first (loss {float(ref_loss):.6f}, |grad| {float(ref.norm()):.4f})")
    bad_total += bad
print("RACE STRESS", "OK" if bad_total == 0 else "FAILED")
sys.exit(1 if bad_total else 0)
