#!/usr/bin/env python3
"""Fixture generator for the CustomTransformer row.  TEST INFRASTRUCTURE, runs only in the build container.

Imports the reference's ``dquartic/model/building_blocks.py`` from /root/reference (torch only, no stand-ins needed) and records
for two small configurations: the module's state_dict (its own default initialisation under ``torch.manual_seed``), seeded
inputs, the forward output, and the autograd gradients of ``sum(out * probe)`` w.r.t. every parameter and both float inputs.
Output: tests/golden/tfm_tiny.npz."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def capture(out, tag, input_dim, hidden, heads, layers, B, S1, S2, seed):
    from dquartic.model.building_blocks import CustomTransformer  # the reference's

    torch.manual_seed(seed)
    net = CustomTransformer(input_dim=input_dim, hidden_dim=hidden, num_heads=heads, num_layers=layers)
    net.train()
    x_t = torch.randn(B, S1, input_dim, requires_grad=True)
    x_cond = torch.randn(B, S2, requires_grad=True)
    t = torch.randint(0, 1000, (B,))
    probe = torch.randn(B, S1, input_dim)
    y = net(x_t, t, x_cond)
    (y * probe).sum().backward()
    out[f"{tag}/config"] = np.asarray([input_dim, hidden, heads, layers, B, S1, S2], np.int64)
    out[f"{tag}/x_t"], out[f"{tag}/x_cond"], out[f"{tag}/t"], out[f"{tag}/probe"] = x_t.detach().numpy(), x_cond.detach().numpy(), t.numpy(), probe.numpy()
    out[f"{tag}/out"] = y.detach().numpy()
    out[f"{tag}/grad/x_t"], out[f"{tag}/grad/x_cond"] = x_t.grad.numpy(), x_cond.grad.numpy()
    for k, v in net.state_dict().items():
        out[f"{tag}/param/{k}"] = v.detach().numpy()
    for k, v in net.named_parameters():
        out[f"{tag}/grad/{k}"] = v.grad.numpy()
    out[f"{tag}/keys"] = np.asarray(list(net.state_dict().keys()))
    # eval-mode forward (nn.MultiheadAttention's fused inference path) must agree with the training-mode one
    net.eval()
    with torch.no_grad():
        out[f"{tag}/out_eval"] = net(x_t.detach(), t, x_cond.detach()).numpy()


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    out = {}
    capture(out, "a", input_dim=24, hidden=16, heads=2, layers=2, B=2, S1=5, S2=3, seed=0)
    capture(out, "b", input_dim=40, hidden=32, heads=4, layers=1, B=3, S1=7, S2=9, seed=1)
    path = os.path.join(REPO, "tests", "golden", "tfm_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
