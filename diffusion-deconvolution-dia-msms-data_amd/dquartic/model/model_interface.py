"""Training / prediction harness -- drop-in for the reference's ``dquartic.model.model_interface``
(reference model_interface.py:64-236 schedulers/callbacks, :238-1150 ModelInterface).

Kept: the public method names, signatures, checkpoint dictionary (``epoch, model_state_dict, optimizer_state_dict,
scheduler_state_dict, best_loss``; reference :617-626), "latest" + "best" checkpoint cadence (:419-430), per-epoch
warm-up + cosine LambdaLR (:149-155, :400), the ``(ms2_1, ms1_1, ms2_2, ms1_2)`` batch contract with
``ms2_cond = 0.5*ms2_1 + 0.5*ms2_2`` (:1070-1075) and the step sequence of ``_train_one_batch`` (:1112-1123):
zero_grad -> train_step -> backward -> clip_grad_norm_(10) -> AdamW -> loss float.

Changed (DESIGN.md "deviations"): when the network is this package's ``UNet1d`` the step runs through two native
calls on flat buffers (``dq_train_step`` + ``dq_adamw_clip_step``) instead of autograd; with ``torch.distributed``
initialised the flat gradient is all-reduced (RCCL) between them; wandb / plotting are optional and imported lazily;
the "latest" checkpoint goes next to ``checkpoint_path`` ('.' when it has no directory, not the filesystem root).
"""
import math
import os
from typing import List

import numpy as np
import torch

from .. import _native as N


# ----------------------------------------------------------------------------------------------------------------
# learning-rate schedule (reference model_interface.py:64-194)
# ----------------------------------------------------------------------------------------------------------------
class LR_SchedulerInterface(object):
    def __init__(self, optimizer: torch.optim.Optimizer, **kwargs):
        raise NotImplementedError

    def step(self, epoch: int, loss: float):
        raise NotImplementedError

    def get_last_lr(self) -> float:
        raise NotImplementedError


class WarmupLR_Scheduler(LR_SchedulerInterface):
    """Linear warm-up then cosine decay, stepped once per epoch."""

    def __init__(self, optimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5, last_epoch: int = -1):
        self.optimizer = optimizer
        self.lambda_lr = self.get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles, last_epoch)

    def step(self, epoch: int = None, loss=None):
        return self.lambda_lr.step()

    def get_last_lr(self) -> List[float]:
        return self.lambda_lr.get_last_lr()

    @staticmethod
    def _lr_lambda(current_step: int, *, num_warmup_steps: int, num_training_steps: int, num_cycles: float):
        if current_step < num_warmup_steps:
            return float(current_step + 1) / float(max(1, num_warmup_steps))
        progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(1e-10, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))

    def get_cosine_schedule_with_warmup(self, optimizer, num_warmup_steps, num_training_steps, num_cycles=0.5, last_epoch=-1):
        from functools import partial

        fn = partial(self._lr_lambda, num_warmup_steps=num_warmup_steps, num_training_steps=num_training_steps, num_cycles=num_cycles)
        return torch.optim.lr_scheduler.LambdaLR(optimizer, fn, last_epoch)


class CallbackHandler:
    """Hooks at epoch / batch end; ``epoch_callback`` returning False stops training (reference :196-236)."""

    def epoch_callback(self, epoch: int, epoch_loss: float) -> bool:
        return True

    def batch_callback(self, batch: int, batch_loss: float):
        pass


# ----------------------------------------------------------------------------------------------------------------
# AdamW over the flat parameter buffer (K11)
# ----------------------------------------------------------------------------------------------------------------
class FlatAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` semantics (torch defaults) executed by ``dq_adamw_clip_step`` on the network's flat
    parameter / gradient buffers, with the global-norm clip of ``clip_grad_norm_`` folded in.  ``state_dict()`` has the
    torch AdamW layout (per-parameter ``step, exp_avg, exp_avg_sq``), so checkpoints interchange with the reference."""

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=10.0):
        self.net = net
        params = [p for _, p in net.trainable_named()]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = max_norm
        self.grad_scale = 1.0
        self._m = self._v = self._scratch = self._gnorm = None
        self._step = 0
        self._step_dev = self._lr_dev = None   # device copies of the step count / learning rate (step_dev(): graph-replayable step)
        self._lr_dev_value = None

    def _buffers(self):
        flat = self.net.flat_params
        if self._m is None or self._m.device != flat.device:
            old_m, old_v = self._m, self._v
            self._m = torch.zeros_like(flat) if old_m is None else old_m.to(flat.device)
            self._v = torch.zeros_like(flat) if old_v is None else old_v.to(flat.device)
            self._scratch = torch.empty(1024, dtype=torch.float32, device=flat.device)
            self._gnorm = torch.zeros((), dtype=torch.float32, device=flat.device)
            self._publish_state()
        return flat

    def _publish_state(self):
        for (name, o, shape), p in zip(self.net._layout, self.param_groups[0]["params"]):
            n = p.numel()
            self.state[p] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[o:o + n].view(shape),
                             "exp_avg_sq": self._v[o:o + n].view(shape)}

    @torch.no_grad()
    def step(self, closure=None):
        flat = self._buffers()
        grads = self.net.flat_grads()
        g = self.param_groups[0]
        self._step += 1
        N.check(N.lib().dq_adamw_clip_step(N.ptr(flat), N.ptr(grads), N.ptr(self._m), N.ptr(self._v), flat.numel(), N.ptr(self._scratch),
                                           float(self.grad_scale), float(self.max_norm), float(g["lr"]), float(g["betas"][0]),
                                           float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), int(self._step),
                                           N.ptr(self._gnorm), N.stream_ptr()), "dq_adamw_clip_step")
        return None  # (the per-parameter "step" entries are refreshed by state_dict(): 395 tensor constructions per step cost ~1 ms of host time)

    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def _dev_state(self):
        """Device copies of the step count and the learning rate for ``step_dev``; the lr copy is refreshed when the scheduler changed it."""
        flat = self._buffers()
        if self._step_dev is None or self._step_dev.device != flat.device:
            self._step_dev = torch.tensor([self._step], dtype=torch.int32, device=flat.device)
            self._lr_dev = torch.zeros(2, dtype=torch.float32, device=flat.device)  # (hi, lo): the double lr as two floats
            self._lr_dev_value = None
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_dev_value:
            hi = float(np.float32(lr))
            self._lr_dev.copy_(torch.tensor([hi, float(np.float32(lr - hi))], dtype=torch.float32), non_blocking=False)
            self._lr_dev_value = lr
        return flat

    @torch.no_grad()
    def step_dev(self, count: bool = True):
        """The same update with the step count and lr read from device memory (``dq_adamw_clip_step_dev``): the call is identical from
        step to step, so it can sit in a captured graph.  ``count=False`` while a graph is being captured (the replay counts instead)."""
        flat = self._dev_state()
        grads = self.net.flat_grads()
        g = self.param_groups[0]
        if count:
            self._step += 1
        N.check(N.lib().dq_adamw_clip_step_dev(N.ptr(flat), N.ptr(grads), N.ptr(self._m), N.ptr(self._v), flat.numel(), N.ptr(self._scratch),
                                               float(self.grad_scale), float(self.max_norm), N.ptr(self._lr_dev), float(g["betas"][0]),
                                               float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), N.ptr(self._step_dev),
                                               N.ptr(self._gnorm), N.stream_ptr()), "dq_adamw_clip_step_dev")
        return None

    def sync_step_dev(self):
        """After the host-side step count changed outside step_dev (plain step(), load_state_dict): push it to the device copy."""
        if self._step_dev is not None:
            self._step_dev.fill_(self._step)

    def zero_grad(self, set_to_none: bool = False):
        self.net.flat_grads(zero=True)

    @property
    def last_grad_norm(self) -> torch.Tensor:
        """Pre-clip global gradient norm of the last step (0-dim device tensor)."""
        return self._gnorm

    def load_state_dict(self, state_dict):
        self._buffers()
        super().load_state_dict(state_dict)
        # copy the loaded moments back into the flat buffers and re-publish views
        step = 0
        for (name, o, shape), p in zip(self.net._layout, self.param_groups[0]["params"]):
            st = self.state.get(p)
            if st:
                n = p.numel()
                self._m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self._v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = int(float(st["step"]))
        self._step = step
        self._publish_state()
        self.sync_step_dev()


class BucketedAllReduce:
    """Gradient all-reduce overlapped with the backward that produces the gradients -- what DistributedDataParallel's reducer does
    for the reference (model_interface.py:953-957), driven here by the backward itself: ``on_bucket(i, offset, count)`` is called
    by CustomTransformer._run_bwd when the kernels writing ``grads[offset:offset+count]`` are on the compute stream; the slice's
    all-reduce (sum) is started at once on a communication stream that waits for exactly that point, so a layer's 50 MB of
    gradients (hidden 1024) cross xGMI while the layers below it are still being differentiated.  ``finish()`` makes the compute
    stream wait for every bucket.  One bucket per layer: at hidden 1024 a bucket is 12.6 M floats, well past the size where a
    ring all-reduce is link-bound rather than latency-bound, so layers are not merged further.

    On CPU tensors (gloo; tests/test_dp_gloo.py) the same calls run without streams."""

    def __init__(self, grads: torch.Tensor, group=None):
        self.grads = grads
        self.group = group
        self.works = []
        self.seen = []
        self.comm = torch.cuda.Stream(device=grads.device) if grads.is_cuda else None

    def on_bucket(self, i: int, offset: int, count: int):
        if offset < 0 or count <= 0 or offset + count > self.grads.numel():
            raise ValueError(f"bucket {i}: slice [{offset}, {offset + count}) outside the {self.grads.numel()}-float gradient buffer")
        piece = self.grads[offset:offset + count]
        self.seen.append((offset, count))
        if self.comm is None:
            self.works.append(torch.distributed.all_reduce(piece, group=self.group, async_op=True))
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.grads.device))
        self.comm.wait_event(ready)
        with torch.cuda.stream(self.comm):
            self.works.append(torch.distributed.all_reduce(piece, group=self.group, async_op=True))

    def finish(self):
        """Wait for every bucket (on CUDA: the current stream waits, the host does not) and check the buckets covered the buffer."""
        for w in self.works:
            w.wait()
        if self.comm is not None:
            torch.cuda.current_stream(self.grads.device).wait_stream(self.comm)
        covered = sorted(self.seen)
        pos = 0
        for off, cnt in covered:
            if off != pos:
                raise RuntimeError(f"gradient buckets leave [{pos}, {off}) unreduced" if off > pos else f"gradient buckets overlap at {off}")
            pos = off + cnt
        if pos != self.grads.numel():
            raise RuntimeError(f"gradient buckets leave [{pos}, {self.grads.numel()}) unreduced")
        self.works, self.seen = [], []


class TrainStepGraph:
    """One optimiser step of ``_train_one_batch`` -- draw t and noise, q_sample, forward, loss, backward, clip, AdamW -- captured in a
    hipGraph (through ``torch.cuda.graph``: the library's launches go to the capture stream like every other caller's stream) and replayed
    with ONE host call per step.  Single-process training only (a gradient all-reduce is not captured).  The inputs are copied into fixed
    buffers before each replay; t and noise come from torch's graph-safe generator, so successive replays draw fresh values; the step
    count and lr are read from device memory by the optimiser kernel.  The backward's remaining weight-gradient launches run on the capture
    stream (``dq_plan_set_side_stream(plan, 0)``): a fork / join inside a graph was measured slower than the chain."""

    def __init__(self, dm, x_0, ms2_cond, ms1_cond, ms1_loss_weight=0.0, keep_side=False):
        self.dm, self.opt, self.net = dm, dm.optimizer, dm.model
        self.key = (tuple(x_0.shape), tuple(ms1_cond.shape), float(ms1_loss_weight or 0.0), float(dm.optimizer.grad_scale))
        self.x0, self.c2, self.c1 = (torch.empty_like(v, dtype=torch.float32).copy_(v) for v in (x_0, ms2_cond, ms1_cond))
        self.w = float(ms1_loss_weight or 0.0)
        self.keep_side = bool(keep_side)  # keep the fork / join inside the captured step (measured slower than the single chain: off)
        N.check(N.lib().dq_plan_set_side_stream(self.net._plan, 1 if self.keep_side else 0), "dq_plan_set_side_stream")
        self.opt._dev_state()
        self.opt.sync_step_dev()
        # warm-up on a side stream (workspaces, occupancy queries, lazy stream creation: nothing may allocate during capture) -- on a
        # SNAPSHOT of the training state: parameters, moments, step count and the generator state are put back afterwards, so that
        # building the graph is not a training step
        snap = (self.net.flat_params.detach().clone(), self.opt._m.clone(), self.opt._v.clone(), self.opt._step, torch.cuda.get_rng_state(x_0.device))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body(count=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            self.net.flat_params.copy_(snap[0]); self.opt._m.copy_(snap[1]); self.opt._v.copy_(snap[2])
        self.opt._step = snap[3]
        self.opt.sync_step_dev()
        torch.cuda.set_rng_state(snap[4], x_0.device)
        # the replay dereferences RAW device pointers: every buffer the captured launches touch is held here for the graph's lifetime (the
        # network's workspace cache may evict its entry when another training shape comes by -- this reference keeps the memory alive), and
        # the ones whose identity carries the training state are part of matches(): a `.to()` / storage-replacing load re-creates the flat
        # buffers, and a graph that still pointed at the old ones would train dead memory
        B, RT = int(x_0.shape[0]), int(x_0.shape[1])
        self._held = self._live_buffers() + (self.net.workspace(B, RT, True), self.opt._scratch, self.opt._gnorm, self.opt._lr_dev,
                                             self.opt._step_dev, dm.alpha_bars)
        self._ptrs = tuple(t.data_ptr() for t in self._live_buffers())
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body(count=False)
        if tuple(t.data_ptr() for t in self._live_buffers()) != self._ptrs or self.net.workspace(B, RT, True) is not self._held[4]:
            raise RuntimeError("TrainStepGraph: a training buffer moved while the step was being captured")

    def _live_buffers(self):
        return (self.net.flat_params, self.net.flat_grads(), self.opt._m, self.opt._v)

    def _body(self, count):
        loss = self.dm.train_step_fused(self.x0, self.c2, self.c1, zero_grads=True, ms1_loss_weight=self.w)
        self.opt.step_dev(count=count)
        return loss

    def matches(self, x_0, ms1_cond, ms1_loss_weight, grad_scale):
        return (self.key == (tuple(x_0.shape), tuple(ms1_cond.shape), float(ms1_loss_weight or 0.0), float(grad_scale))
                and tuple(t.data_ptr() for t in self._live_buffers()) == self._ptrs)

    def step(self, x_0, ms2_cond, ms1_cond):
        self.x0.copy_(x_0); self.c2.copy_(ms2_cond); self.c1.copy_(ms1_cond)
        self.opt._dev_state()      # (a changed lr reaches its device copy here, outside the graph)
        self.opt._step += 1
        self.graph.replay()
        return self.loss.clone()   # a fresh tensor per step, as the eager path returns (the graph's own output is overwritten by the next replay)


# ----------------------------------------------------------------------------------------------------------------
# harness
# ----------------------------------------------------------------------------------------------------------------
class ModelInterface(object):
    def __init__(self, device: str = torch.device("cuda" if torch.cuda.is_available() else "cpu"), min_pred_value: float = 0.0, **kwargs):
        self.model: torch.nn.Module = None
        self.optimizer = None
        self.model_params: dict = {}
        self.min_pred_value = min_pred_value
        self.lr_scheduler_class = WarmupLR_Scheduler
        self.callback_handler = CallbackHandler()
        self.device = device
        self.ms1_loss_weight = None
        self.use_wandb = False
        self.last_grad_norm = None

    def __repr__(self):
        return f"{self.__class__.__name__} with {self.model.__class__.__name__} model with {self.get_parameter_num()} parameters on {self.device}"

    # ---- public
    def build(self, model_class, **kwargs):
        self.model = model_class
        self._init_for_training()

    def get_parameter_num(self):
        return int(np.sum([p.numel() for p in self.model.parameters()]))

    def train_step(self, x_0, ms2_cond=None, ms1_cond=None, noise=None, ms1_loss_weight=0.0):
        raise NotImplementedError

    def sample(self, x_t, ms2_cond=None, ms1_cond=None, num_steps=1000):
        raise NotImplementedError

    def train(self, dataloader, batch_size, epochs, warmup_epochs: int = 5, learning_rate: float = 1e-4, use_wandb: bool = False,
              checkpoint_path: str = "best_model.ckpt", **kwargs):
        """Epoch loop (reference :453-559).  ``warmup_epochs > 0`` uses the warm-up/cosine schedule, else a constant lr.
        ``batch_size`` is accepted and unused, as in the reference (the dataloader carries it)."""
        if warmup_epochs > 0:
            self.train_with_warmup(dataloader, epochs, num_warmup_steps=int(warmup_epochs), learning_rate=learning_rate,
                                   use_wandb=use_wandb, checkpoint_path=checkpoint_path, **kwargs)
        else:
            self.train_with_warmup(dataloader, epochs, num_warmup_steps=0, learning_rate=learning_rate, use_wandb=use_wandb,
                                   checkpoint_path=checkpoint_path, constant_lr=True, **kwargs)

    def train_with_warmup(self, dataloader, num_epochs, num_warmup_steps=5, learning_rate=1e-4, use_wandb=True,
                          log_every_n_epochs=100, checkpoint_path="best_model.ckpt", *, constant_lr: bool = False, **kwargs):
        """Reference :348-450, same positional order (dataloader, num_epochs, num_warmup_steps, learning_rate, use_wandb,
        log_every_n_epochs, checkpoint_path).  ``constant_lr`` (keyword-only, this build) is how ``train`` runs its
        ``warmup_epochs <= 0`` branch (reference :498-559) through the same loop."""
        self.use_wandb_epoch = bool(use_wandb)
        wandb = _wandb() if use_wandb else None
        self._prepare_training(learning_rate)
        lr_scheduler = None if constant_lr else self._get_lr_schedule_with_warmup(num_warmup_steps, num_epochs)
        self.model.train()
        ckpt_dir = os.path.dirname(checkpoint_path) or "."
        latest = os.path.join(ckpt_dir, "dquartic_latest_checkpoint.ckpt")
        start_epoch, best_loss, lr_scheduler = self.load_checkpoint(lr_scheduler, latest, self.device)
        self._sync_replicas()  # data-parallel: every rank continues from rank 0's weights / moments (also after a resume)
        # ... and from rank 0's epoch counter, best loss, lr-schedule state and current lr: with the "latest" checkpoint visible to rank 0
        # only (node-local disks), the ranks would otherwise run epoch loops of different lengths and hang in the per-step all-reduce
        start_epoch, best_loss = self._sync_resume_state(start_epoch, best_loss, lr_scheduler)
        best_epoch = start_epoch
        rank0 = _rank() == 0
        for epoch in range(start_epoch, num_epochs):
            if hasattr(getattr(dataloader, "sampler", None), "set_epoch"):
                dataloader.sampler.set_epoch(epoch)  # DistributedSampler: a different permutation every epoch
            if hasattr(dataloader.dataset, "reset_epoch"):
                dataloader.dataset.reset_epoch()
            batch_loss = self._train_one_epoch(epoch, dataloader)
            if lr_scheduler is not None:
                lr_scheduler.step(epoch, np.mean(batch_loss))
            avg = self._global_mean(float(np.mean(batch_loss)))  # DP: the mean over ranks (one float, logging only)
            lr_now = self.optimizer.param_groups[0]["lr"]
            if wandb is not None and rank0:
                wandb.log({"epoch": epoch, "train/loss": avg, "learning_rate": lr_now})
            if rank0:
                print(f"[Training] Epoch={epoch + 1}, lr={lr_now}, loss={avg}")
                self.save_checkpoint(lr_scheduler, epoch, avg, latest)
                if avg < best_loss:
                    best_loss, best_epoch = avg, epoch + 1
                    self.save_checkpoint(lr_scheduler, epoch, best_loss, checkpoint_path)
            elif avg < best_loss:
                best_loss, best_epoch = avg, epoch + 1
            if not self.callback_handler.epoch_callback(epoch=epoch, epoch_loss=avg):
                print(f"Training stopped at epoch {epoch}")
                break
        if rank0:
            print(f"Best model checkpoint saved at epoch {best_epoch} with loss: {best_loss:.6f}")

    # ---- data-parallel helpers (new work: the reference is single-process)
    def _sync_replicas(self, src: int = 0):
        """Make every rank's replica identical to rank ``src``'s: the flat parameter buffer and, when an optimiser exists, its
        AdamW moments and step count.  Each process builds its network from its own default-seeded RNG, and only gradients are
        exchanged per step, so without this the replicas would start (or resume) from different weights and never meet."""
        d = torch.distributed
        if not (d.is_available() and d.is_initialized()) or d.get_world_size() == 1:
            return
        if hasattr(self.model, "flat_params"):
            d.broadcast(self.model.flat_params, src=src)
            for b in self.model.buffers():
                d.broadcast(b, src=src)
            if isinstance(self.optimizer, FlatAdamW):
                self.optimizer._buffers()
                d.broadcast(self.optimizer._m, src=src)
                d.broadcast(self.optimizer._v, src=src)
                step = torch.tensor([float(self.optimizer._step)], device=self.optimizer._m.device)
                d.broadcast(step, src=src)
                self.optimizer._step = int(step.item())
        else:
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                d.broadcast(t.data, src=src)

    def _sync_resume_state(self, start_epoch, best_loss, lr_scheduler, src: int = 0):
        """Rank ``src``'s (start_epoch, best_loss), LambdaLR state and per-group lr on every rank (no-op outside a process group)."""
        d = torch.distributed
        if not (d.is_available() and d.is_initialized()) or d.get_world_size() == 1:
            return start_epoch, best_loss
        mine = d.get_rank() == src
        box = [{"start_epoch": start_epoch, "best_loss": best_loss,
                "scheduler": lr_scheduler.lambda_lr.state_dict() if lr_scheduler is not None else None,
                "lrs": [g["lr"] for g in self.optimizer.param_groups] if self.optimizer is not None else None} if mine else None]
        dev = None
        if d.get_backend() == "nccl":  # (object collectives stage through the current device with RCCL)
            dev = self.model.flat_params.device if hasattr(self.model, "flat_params") else next(self.model.parameters()).device
        d.broadcast_object_list(box, src=src, device=dev)
        st = box[0]
        if not mine:
            if lr_scheduler is not None and st["scheduler"] is not None:
                lr_scheduler.lambda_lr.load_state_dict(st["scheduler"])
            if self.optimizer is not None and st["lrs"] is not None:
                for g, lr in zip(self.optimizer.param_groups, st["lrs"]):
                    g["lr"] = lr
        return st["start_epoch"], st["best_loss"]

    def _global_mean(self, value: float) -> float:
        """Mean over the ranks of a per-rank scalar (the epoch's mean loss): identical on every rank afterwards, so the
        "best" decision, the callback and the log line agree."""
        d = torch.distributed
        if not (d.is_available() and d.is_initialized()) or d.get_world_size() == 1:
            return value
        dev = self.model.flat_params.device if hasattr(self.model, "flat_params") else next(self.model.parameters()).device
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        d.all_reduce(t)
        return float(t.item()) / d.get_world_size()

    def load_checkpoint(self, scheduler, checkpoint_path, device):
        if os.path.exists(checkpoint_path):
            print(f"Loading checkpoint from {checkpoint_path}...")
            ck = torch.load(checkpoint_path, map_location=device, weights_only=False)
            self.model.load_state_dict(ck["model_state_dict"])
            if self.optimizer is not None and ck.get("optimizer_state_dict") is not None:
                self.optimizer.load_state_dict(ck["optimizer_state_dict"])
            if scheduler is not None and ck.get("scheduler_state_dict") is not None:
                scheduler.lambda_lr.load_state_dict(ck["scheduler_state_dict"])
            epoch, best_loss = ck["epoch"], ck["best_loss"]
            print(f"Resumed from ({checkpoint_path}) epoch {epoch}, best loss {best_loss:.6f}")
        else:
            print(f"No checkpoint ({checkpoint_path}) found. Starting from scratch.")
            epoch, best_loss = 0, float("inf")
        return epoch, best_loss, scheduler

    def save_checkpoint(self, scheduler, epoch, best_loss, checkpoint_path):
        torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                    "scheduler_state_dict": (scheduler.lambda_lr.state_dict() if scheduler is not None else None),
                    "best_loss": best_loss}, checkpoint_path)

    def predict(self, dataloader, mixture_weights=(0.5, 0.5), num_steps=1000):
        """Reference :630-668: one dict per batch with the first item's prediction (``_predict_one_batch`` returns item 0)."""
        self.model.eval()
        preds = []
        for ms2_1, ms1_1, ms2_2, ms1_2 in dataloader:
            x_0, ms1_cond = ms2_1.to(self.device), ms1_1.to(self.device)
            ms2_cond = (ms2_1 * mixture_weights[0]).to(self.device) + (ms2_2 * mixture_weights[1]).to(self.device)
            pred, _ = self._predict_one_batch(x_0, ms2_cond=ms2_cond, ms1_cond=ms1_cond, num_steps=num_steps)
            preds.append({"ms2_1": ms2_1.cpu().numpy(), "ms1_1": ms1_1.cpu().numpy(), "mixture": ms2_cond.cpu().numpy(), "pred": pred})
        return np.array(preds, dtype=object)

    # ---- internals
    def _init_for_training(self):
        self.loss_func = torch.nn.MSELoss()

    def _prepare_training(self, lr: float, **kwargs):
        self.model.train()
        self._set_lr(lr)
        self._sync_replicas()

    def _native_net(self) -> bool:
        from .building_blocks import DDIMTransformerAdapter
        from .unet1d import UNet1d

        return isinstance(self.model, (UNet1d, DDIMTransformerAdapter))

    def _set_optimizer(self, lr):
        """Reference :1011: AdamW(model.parameters(), lr) with torch defaults."""
        if self._native_net():
            self.optimizer = FlatAdamW(self.model, lr=lr)
        else:
            self.optimizer = torch.optim.AdamW(self.model.parameters(), lr=lr)

    def _set_lr(self, lr: float):
        if self.optimizer is None:
            self._set_optimizer(lr)
        else:
            for g in self.optimizer.param_groups:
                g["lr"] = lr

    def _get_lr_schedule_with_warmup(self, warmup_epoch, epoch):
        if warmup_epoch > epoch:
            warmup_epoch = epoch // 2
        return self.lr_scheduler_class(self.optimizer, num_warmup_steps=warmup_epoch, num_training_steps=epoch)

    def _train_one_epoch(self, epoch, dataloader, mixture_weights=(0.5, 0.5)):
        self.model.train()
        batch_loss = []
        for batch_idx, batch in enumerate(dataloader):
            ms2_1, ms1_1, ms2_2, ms1_2 = batch
            x_0, ms1_cond = ms2_1.to(self.device), ms1_1.to(self.device)
            # simulated mixed spectra from the target window and the other window (reference :1073-1075); a resident loader
            # (utils/data_loader.py:ResidentPairLoader) has already formed it on the GPU in the kernel that normalised the pair
            if getattr(batch, "ms2_cond", None) is not None and batch.mixture_weights == tuple(float(w) for w in mixture_weights):
                ms2_cond = batch.ms2_cond
            else:
                ms2_cond = (ms2_1 * mixture_weights[0]).to(self.device) + (ms2_2 * mixture_weights[1]).to(self.device)
            loss = self._train_one_batch(x_0, ms2_cond=ms2_cond, ms1_cond=ms1_cond, noise=None, ms1_loss_weight=self.ms1_loss_weight)
            batch_loss.append(loss)
            self.callback_handler.batch_callback(batch_idx, loss)
        return batch_loss

    def enable_train_graph(self, on: bool = True, keep_side: bool = False):
        """Run ``_train_one_batch`` as one captured hipGraph replay per step (``TrainStepGraph``): the reference trains at batch_size 1
        (dquartic_train_config.json:12), where a step is a chain of ~250 short launches and the host's launch rate is the limit.  Same
        kernels in the same order as the eager step (bit-identical given the same t / noise); single-process runs with drawn t / noise."""
        self.train_graph = bool(on)
        if bool(keep_side) != getattr(self, "_train_graph_side", False):
            self._train_graphs = {}
        self._train_graph_side = bool(keep_side)  # capture the side-stream fork / join as a second branch of the graph
        if not on and getattr(self, "_train_graphs", None):
            self._train_graphs = {}
            N.check(N.lib().dq_plan_set_side_stream(self.model._plan, 1), "dq_plan_set_side_stream")  # eager steps fork their weight gradients again

    def _train_one_batch(self, x_0, ms2_cond=None, ms1_cond=None, noise=None, ms1_loss_weight=0.0, t=None, sync=True):
        """Reference :1090-1123.  Returns the loss as a float (``sync=False``: a 0-dim device tensor, no host sync)."""
        fused = self._native_net() and isinstance(self.optimizer, FlatAdamW) and x_0.is_cuda and hasattr(self, "train_step_fused")
        if (fused and getattr(self, "train_graph", False) and noise is None and t is None and _world() == 1 and hasattr(self.model, "_plan")
                and ms2_cond is not None and ms1_cond is not None):
            # the whole step as one captured graph (enable_train_graph): drawn t / noise, single process.  A few graphs are kept, keyed by
            # shape: an epoch's short last batch does not throw the full-batch graph away (a capture costs two warm-up steps + state copies)
            graphs = self.__dict__.setdefault("_train_graphs", {})
            key = (tuple(x_0.shape), tuple(ms1_cond.shape), float(ms1_loss_weight or 0.0))
            tg = graphs.get(key)
            if tg is None or not tg.matches(x_0, ms1_cond, ms1_loss_weight, 1.0):
                for k in [k for k, g in graphs.items() if not g.matches(g.x0, g.c1, g.w, 1.0)]:
                    del graphs[k]   # (its flat buffers were re-created since: it can never match again)
                while len(graphs) >= 4:
                    del graphs[next(iter(graphs))]
                self.optimizer.grad_scale = 1.0
                tg = graphs[key] = TrainStepGraph(self, x_0, ms2_cond, ms1_cond, ms1_loss_weight, keep_side=getattr(self, "_train_graph_side", False))
            loss = tg.step(x_0, ms2_cond, ms1_cond)
            self.last_grad_norm = self.optimizer.last_grad_norm
            return loss.item() if sync else loss
        if fused:
            if noise is not None:
                noise = self.normalize(noise)  # reference quirk: a passed noise is mapped 2n-1 (model.py:346)
            loss = self.train_step_fused(x_0, ms2_cond, ms1_cond, t=t, noise=noise, zero_grads=True, ms1_loss_weight=ms1_loss_weight or 0.0)
            world = _world()
            if getattr(self, "_grads_reduced", False):
                self._grads_reduced = False  # the transformer's backward already all-reduced its buckets (BucketedAllReduce)
            elif torch.distributed.is_available() and torch.distributed.is_initialized():
                torch.distributed.all_reduce(self.model.flat_grads())  # one flat RCCL all-reduce (sum) of 515 KB
            self.optimizer.grad_scale = 1.0 / world
            self.optimizer.step()
            self.optimizer.sync_step_dev() if self.optimizer._step_dev is not None else None
            self.last_grad_norm = self.optimizer.last_grad_norm
            return loss.item() if sync else loss
        self.optimizer.zero_grad()
        loss = self.train_step(x_0, ms2_cond=ms2_cond, ms1_cond=ms1_cond, noise=noise, ms1_loss_weight=ms1_loss_weight or 0.0)
        if loss.dim() > 0:
            loss = loss.mean()
        loss.backward()
        if isinstance(self.optimizer, FlatAdamW):
            self.optimizer.grad_scale = 1.0
            self.optimizer.step()  # clip folded in
            self.last_grad_norm = self.optimizer.last_grad_norm
        else:
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=10.0)
            self.optimizer.step()
        return loss.item() if sync else loss.detach()

    def _predict_one_batch(self, x_0, ms2_cond=None, ms1_cond=None, num_steps=1000):
        """Reference :1125-1150: eval + no_grad + sample(randn_like(x_0)); returns item 0 of the batch as numpy."""
        self.model.eval()
        with torch.no_grad():
            sample, pred_noise = self.sample(torch.randn_like(x_0), ms2_cond=ms2_cond, ms1_cond=ms1_cond, num_steps=num_steps)
        return sample[0].cpu().detach().numpy(), pred_noise[0].cpu().detach().numpy()

    def log_single_prediction(self, *args, **kwargs):
        raise NotImplementedError("wandb prediction tables / pyopenms_viz plots are outside the hot path (SURVEY section 2)")

    plot_single_prediction = log_single_prediction


def _wandb():
    try:
        import wandb

        return wandb
    except ImportError:
        print("wandb is not installed; continuing without it")
        return None


def _world() -> int:
    d = torch.distributed
    return d.get_world_size() if d.is_available() and d.is_initialized() else 1


def _rank() -> int:
    d = torch.distributed
    return d.get_rank() if d.is_available() and d.is_initialized() else 0
