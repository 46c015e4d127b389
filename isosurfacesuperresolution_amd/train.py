"""Training step of the unshaded temporal network, and its data-parallel form.

Restates ``trainNormal`` of ``SuperresolutionNetwork/mainVideoUnshaded.py:397-473`` (optimizer and
scheduler of ``:287-300``: Adam(lr 1e-4), ``StepLR(lrStep, 0.5)`` stepped at epoch start; seeds of
``:170-171``):

    for each clip [B, T, ...]:   loss = 0, previous_output = None
      for j in range(T):
        j == 0: previous_warped = initialImage(input[:,0], 6, mode);  previous_warped_loss = target[:,0]
        j  > 0: previous_warped = warp_upscale(previous_output, flow[:,j-1], 4, special_mask=True)
        prediction = model(cat(input[:,j], flatten_high(previous_warped, 4)))
        loss += criterion(target[:,j], prediction, up(input[:,j]), previous_input, previous_warped_loss)
        previous_output = cat(clamp(mask), normalize(normal), clamp(depth), clamp(ao))   # NOT detached
      loss.backward(); optimizer.step()

The reference has nothing distributed.  ``DataParallelTrainer`` adds the
MI355X form: one process per GPU, the global batch split over ranks, identical initial weights
(rank 0 broadcast), and after ``backward`` ONE all-reduce of a single flat 3.64 MB gradient bucket
(911 046 fp32; RCCL over xGMI is latency bound at this size, so per-layer buckets would only add
launches), averaged, then a local Adam step.  Loss terms are batch means, so averaged gradients
equal the single-process gradients of the global batch.
"""
import math
import os
import warnings

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops
from .models import VideoTools
from .utils import ScreenSpaceShading, initialImage


def backward(loss):
    """``loss.backward()`` with the weight gradients of the HIP convolutions deferred to ONE pass per layer over all
    frames of the clip (``ops.deferred_weight_gradients``) instead of one pass per layer and frame.

    Limitation: a deferred gradient is added to ``param.grad`` when the context exits and does not pass through
    autograd's accumulation, so hooks registered on a convolution weight would not see it -- parameters that carry
    tensor hooks or post-accumulate-grad hooks are detected and keep the ordinary per-frame path
    (``ops._weight_grad_or_defer``).  ``DataParallelTrainer`` reduces ``param.grad`` after this call and is unaffected."""
    with ops.deferred_weight_gradients():
        loss.backward()


def clip_loss(model, criterion, input, flow, target, initial_image="zero", upscale=4, upsample="bilinear",
              disable_temporal=False, fused_recurrence=True):
    """input [B,T,5,h,w], flow [B,T,2,h,w], target [B,T,6,4h,4w] -> (loss tensor, sum of the per-frame losses).
    The reference reads every frame's loss back with ``.item()`` (mainVideoUnshaded.py:454); here the sum is accumulated
    on the device and read back ONCE after the last frame, so the whole clip is enqueued without a stall."""
    B, T, Cout, Hh, Wh = target.shape
    if ops.TRAIN_SPLIT and not ops.TRAIN_BF16 and target.is_cuda and torch.is_grad_enabled():
        # every weight changed in the last optimizer step: refresh all split-kernel weight images in two launches
        ops.prepare_split_many([m.weight for m in model.modules()
                                if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.out_channels > 8 and m.in_channels > 8])
    prediction = None
    loss = 0
    loss_sum = None
    lazy_before = getattr(criterion, 'lazy_values', None)
    if lazy_before is not None:
        criterion.lazy_values = True
    kw = dict(mode=upsample, **({"align_corners": False} if upsample in ("bilinear", "bicubic") else {}))
    # the reference upsamples (and warps) the low-resolution inputs for every frame and hands them to the criterion
    # (mainVideoUnshaded.py:432-452); LossNetUnshaded never reads them (only the GAN terms would), so a criterion
    # that says so (uses_input = False) is not fed
    feed_inputs = getattr(criterion, 'uses_input', True)
    previous_input = input_high = None
    for j in range(T):
        if j == 0 or disable_temporal:
            previous_warped = initialImage(input[:, 0], Cout, initial_image, False, upscale)
            previous_warped_loss = target[:, 0]
            if feed_inputs:
                previous_input = F.interpolate(input[:, 0], size=(Hh, Wh), **kw)
            single_input = torch.cat((input[:, j], VideoTools.flatten_high(previous_warped, upscale)), dim=1)
        else:
            if fused_recurrence and ops.recurrent_input_supported(prediction, input[:, j], flow[:, j - 1], upscale):
                # clamp / normalize of the previous prediction, warp, space-to-depth and cat in one kernel
                single_input, previous_warped = ops.recurrent_input(prediction, input[:, j], flow[:, j - 1])
            else:
                previous_output = torch.cat([
                    torch.clamp(prediction[:, 0:1], -1, +1),
                    ScreenSpaceShading.normalize(prediction[:, 1:4], dim=1),
                    torch.clamp(prediction[:, 4:5], 0, +1),
                    torch.clamp(prediction[:, 5:6], 0, +1)], dim=1)       # NOT detached: gradients flow through time
                previous_warped = VideoTools.warp_upscale(previous_output, flow[:, j - 1], upscale, special_mask=True)
                single_input = torch.cat((input[:, j], VideoTools.flatten_high(previous_warped, upscale)), dim=1)
            previous_warped_loss = previous_warped
            if feed_inputs:
                previous_input = F.interpolate(input[:, j - 1], size=(Hh, Wh), **kw)
                previous_input = VideoTools.warp_upscale(previous_input, flow[:, j - 1], upscale, special_mask=True)
        prediction, _ = model(single_input)
        if feed_inputs:
            input_high = F.interpolate(input[:, j], size=(Hh, Wh), **kw)
        loss0, _ = criterion(target[:, j], prediction, input_high, previous_input, previous_warped_loss)
        loss = loss + loss0
        loss_sum = loss0.detach() if loss_sum is None else loss_sum + loss0.detach()
    if lazy_before is not None:
        criterion.lazy_values = lazy_before
    return loss, loss_sum


def make_optimizer(model, lr=1e-4, lr_step=500, lr_gamma=0.5, capturable=False, tensor_lr=False, flat=False):
    """Adam + StepLR of mainVideoUnshaded.py:287-300.  ``tensor_lr``: the learning rate lives in a device tensor, so that
    a scheduler step is seen by an optimizer step captured in a HIP graph (needs ``capturable``).  ``flat``: ``FlatAdam`` -- the
    same update over one flat buffer in one launch (parameters and gradients become views of it)."""
    params = list(model.parameters())
    if flat:
        opt = FlatAdam(params, lr=lr, tensor_lr=tensor_lr)
        return opt, torch.optim.lr_scheduler.StepLR(opt, lr_step, lr_gamma)
    if tensor_lr:
        lr = torch.tensor(float(lr), dtype=torch.float32, device=params[0].device)
    opt = torch.optim.Adam(params, lr=lr, capturable=capturable)
    sched = torch.optim.lr_scheduler.StepLR(opt, lr_step, lr_gamma)
    return opt, sched


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (the reference's optimizer, mainVideoUnshaded.py:287-289: lr 1e-4, betas (0.9, 0.999), eps 1e-8, no
    weight decay) over ONE flat buffer: every parameter of the model becomes a view of ``self.flat`` and every gradient a view of
    ``self.bucket``, so a step is one launch (``isrAdamFlatStep``) instead of ~100 per-tensor launches, zeroing the gradients one
    ``zero_``, and the data-parallel exchange one all-reduce of ``self.bucket`` (``DataParallelTrainer`` adopts it).  The step
    counter and (optionally) the learning rate live on the device, so a HIP graph of the step replays correctly.

    Build it AFTER ``model.to(device)`` (moving the model afterwards detaches the views: call ``rebind()``).  A step writes the
    parameters' memory without advancing ``Parameter._version``: it declares the cached kernel-layout weight images stale itself.

    State: the moments and the step counter live in ``self.state[self.flat]`` under torch.optim.Adam's keys (``step``, ``exp_avg``,
    ``exp_avg_sq``), so ``state_dict()`` / ``load_state_dict()`` and pickling the optimizer object -- what the reference's
    checkpoints do (mainVideoUnshaded.py:799-811) -- carry them; the member parameters travel with the pickle and are re-pointed
    at the flat buffers when it is loaded (``__setstate__`` -> ``rebind``).  ``load_state_dict`` copies INTO the existing state
    tensors, so a HIP graph captured before it keeps updating the live ones."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, tensor_lr=False):
        params = [p for p in params if p.requires_grad]
        dev = params[0].device
        self.members = params
        self.flat = self.bucket = None
        flat = self._bind()
        if tensor_lr and not torch.is_tensor(lr):
            lr = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        super().__init__([flat], dict(lr=lr, betas=betas, eps=eps, capturable=True))
        self.state[flat] = dict(step=torch.zeros(1, dtype=torch.float32, device=dev), exp_avg=torch.zeros_like(flat.data),
                                exp_avg_sq=torch.zeros_like(flat.data))

    def _bind(self):
        """(Re)build the flat parameter / gradient buffers from the members' current values and point the members at them."""
        params = self.members
        n = sum(p.numel() for p in params)
        dev, dt = params[0].device, params[0].dtype
        flat = torch.empty(n, dtype=dt, device=dev)
        bucket = torch.zeros(n, dtype=dt, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + k].view_as(p)
                p.grad = bucket[off:off + k].view_as(p)
                off += k
        self.flat = torch.nn.Parameter(flat)
        self.flat.grad = bucket
        self.bucket = bucket
        return self.flat

    # the moments under their old attribute names (kernel launch, tests)
    exp_avg = property(lambda self: self.state[self.flat]['exp_avg'])
    exp_avg_sq = property(lambda self: self.state[self.flat]['exp_avg_sq'])
    steps = property(lambda self: self.state[self.flat]['step'])

    def views_intact(self):
        if self.flat is None or self.bucket is None or self.flat.grad is not self.bucket:
            return False
        base, pbase, esz, off = self.bucket.data_ptr(), self.flat.data_ptr(), self.bucket.element_size(), 0
        for p in self.members:
            if p.grad is None or p.grad.data_ptr() != base + off * esz or p.data_ptr() != pbase + off * esz or p.device != self.flat.device:
                return False
            off += p.numel()
        return True

    def rebind(self):
        """After the members moved (``model.to(device)``) or the optimizer was unpickled: make parameters and gradients views of
        the flat buffers again, keeping the parameters' CURRENT values and the optimizer's moments / step count / learning rate."""
        if self.views_intact():
            return self
        old = self.flat
        st = self.state.pop(old) if old is not None and old in self.state else None
        flat = self._bind()
        self.param_groups[0]['params'] = [flat]
        if st is None:
            st = dict(step=torch.zeros(1), exp_avg=torch.zeros(flat.numel()), exp_avg_sq=torch.zeros(flat.numel()))
        self.state[flat] = {k: (v.to(device=flat.device, dtype=torch.float32 if k == 'step' else flat.dtype) if torch.is_tensor(v) else v)
                            for k, v in st.items()}
        g = self.param_groups[0]
        if torch.is_tensor(g['lr']) and g['lr'].device != flat.device:
            g['lr'] = g['lr'].to(flat.device)
        return self

    def __getstate__(self):
        st = dict(super().__getstate__())
        st['members'] = self.members
        return st

    def __setstate__(self, state):
        members = state.pop('members', None) if isinstance(state, dict) else None
        super().__setstate__(state)
        if members is not None:                    # unpickled (load_state_dict comes through here too, without members)
            self.members = members
            self.flat = self.param_groups[0]['params'][0]
            self.bucket = None                     # gradients are not pickled
            self.rebind()

    def load_state_dict(self, state_dict):
        live = self.state[self.flat]
        lr_live = self.param_groups[0]['lr']
        super().load_state_dict(state_dict)
        loaded = self.state[self.flat]
        with torch.no_grad():
            for k in ('step', 'exp_avg', 'exp_avg_sq'):
                live[k].copy_(loaded[k].reshape(live[k].shape))
        self.state[self.flat] = live
        if torch.is_tensor(lr_live):               # a captured step reads THIS tensor
            lr_live.fill_(float(self.param_groups[0]['lr']))
            self.param_groups[0]['lr'] = lr_live

    def zero_grad(self, set_to_none=False):
        self.bucket.zero_()

    def check_views(self):
        if not self.views_intact():
            raise RuntimeError("FlatAdam: a parameter or its .grad is no longer a view of the flat buffers "
                               "(model.to() after construction -- call rebind() -- or an optimizer.zero_grad(set_to_none=True) elsewhere)")

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        ops.adam_flat_step(self.flat.data, self.bucket, self.exp_avg, self.exp_avg_sq, self.steps, g['lr'], g['betas'][0], g['betas'][1], g['eps'])
        ops.invalidate_weight_images()


def step_scheduler(optimizer, scheduler):
    """``scheduler.step()`` at the START of an epoch, as the reference does (mainVideoUnshaded.py:399; torch >= 1.1 warns
    about the order, the schedule is the reference's).  A learning rate kept in a device tensor is updated IN PLACE:
    a captured optimizer step reads that very tensor."""
    before = [g['lr'] for g in optimizer.param_groups]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scheduler.step()
    for g, old in zip(optimizer.param_groups, before):
        if isinstance(old, torch.Tensor) and g['lr'] is not old:
            old.fill_(float(g['lr']))
            g['lr'] = old
    return float(optimizer.param_groups[0]['lr'])


def train_step(model, criterion, optimizer, batch, **kw):
    """One optimisation step on a clip batch (single process)."""
    input, flow, target = batch
    optimizer.zero_grad()          # (FlatAdam: one zero_ of its flat gradient buffer, the views stay)
    loss, loss_sum = clip_loss(model, criterion, input, flow, target, **kw)
    backward(loss)
    optimizer.step()
    return float(loss_sum.item()) / target.shape[1]


def _optimizer_state_tensors(optimizer):
    for group in optimizer.param_groups:
        for p in group['params']:
            st = optimizer.state.get(p)
            if st:
                for k, v in st.items():
                    if torch.is_tensor(v):
                        yield (id(p), k), v


def _snapshot_training_state(model, optimizer):
    """Copies of everything an optimisation step changes: the model's parameters and buffers, the optimizer's state tensors."""
    return ({k: v.detach().clone() for k, v in model.state_dict().items()},
            {key: v.detach().clone() for key, v in _optimizer_state_tensors(optimizer)})


@torch.no_grad()
def _restore_training_state(model, optimizer, snapshot):
    """Put model and optimizer back IN PLACE (tensors keep their addresses); optimizer state that did not exist at the
    snapshot (a fresh optimizer creates it in its first step) is zeroed, which is what a first step starts from."""
    weights, opt = snapshot
    for k, v in model.state_dict().items():
        v.copy_(weights[k])
    for key, v in _optimizer_state_tensors(optimizer):
        if key in opt:
            v.copy_(opt[key])
        else:
            v.zero_()


class GraphedTrainStep:
    """The whole optimisation step (T frames forward with the recurrence, backward through time, Adam) captured ONCE in
    a HIP graph and replayed: at the per-GPU batch of BASELINE config #3 (2 clips) a step is ~1500 small launches and the
    eager loop is bound by Python, not by the GPU.  Needs a GPU, fixed batch shapes, an optimizer created with
    ``capturable=True`` and (as everywhere in this package) no host read-back inside the step; the loss of each step is
    returned as a device tensor.  ``all_reduce`` (optional callable) runs between backward and the optimizer step --
    ``DataParallelTrainer`` passes its flat-bucket RCCL all-reduce, which is captured with the rest.  ``zero_grad``
    (optional callable) replaces ``optimizer.zero_grad(set_to_none=True)`` -- the data-parallel trainer's gradients are
    views of its flat bucket and must stay those tensors.

    A replay changes the weights WITHOUT advancing ``Parameter._version`` (the captured optimizer kernels write the
    parameters' memory directly), so every replay declares the cached kernel-layout weight images stale
    (``ops.invalidate_weight_images``): inference or an eager step after a replay re-prepares them from the current
    weights.  The graph itself is unaffected: it refreshes its own image buffers at the start of every replay."""

    def __init__(self, model, criterion, optimizer, example_batch, all_reduce=None, zero_grad=None, warmup=3, **kw):
        self.model, self.criterion, self.optimizer, self.kw = model, criterion, optimizer, kw
        self.all_reduce = all_reduce
        if zero_grad is None:
            zero_grad = self.optimizer.zero_grad if isinstance(self.optimizer, FlatAdam) else (lambda: self.optimizer.zero_grad(set_to_none=True))
        self.zero_grad = zero_grad
        self.static = tuple(torch.empty_like(t) for t in example_batch)
        for dst, src in zip(self.static, example_batch):
            dst.copy_(src)
        # The warm-up steps are REAL optimisation steps on the example batch (they create the optimizer's state tensors and
        # every cached buffer the capture must find in place); the reference takes one step per batch (trainNormal,
        # mainVideoUnshaded.py:397-473), so model and optimizer are put back to where they were -- in place, the graph
        # holds these very tensors -- before the first replay.
        before = _snapshot_training_state(model, optimizer) if warmup > 0 else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # PyTorch's capture recipe: warm up on a side stream
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.zero_grad()
        with ops.graph_capture(self.graph):
            self.loss = self._eager()
        if before is not None:
            _restore_training_state(model, optimizer, before)
            self.zero_grad()
        ops.invalidate_weight_images()

    def _eager(self):
        input, flow, target = self.static
        self.zero_grad()
        loss, loss_sum = clip_loss(self.model, self.criterion, input, flow, target, **self.kw)
        backward(loss)
        if self.all_reduce is not None:
            self.all_reduce()
        self.optimizer.step()
        return loss_sum / target.shape[1]

    def __call__(self, batch):
        for dst, src in zip(self.static, batch):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        ops.invalidate_weight_images()
        return self.loss


class DataParallelTrainer:
    """Data-parallel ``train_step``: one process per GPU, one flat gradient all-reduce per step.

    Every parameter's ``.grad`` IS a view of the flat bucket (set once here): backward accumulates straight into it
    (autograd adds in place into an existing ``.grad``; the deferred weight-gradient pass does ``grad.add_``), so the
    exchange is ONE ``all_reduce`` + ONE ``div_`` and zeroing the gradients ONE ``zero_`` -- no per-parameter copies
    around a 42 us collective.  Nothing may replace ``param.grad`` afterwards: use ``trainer.zero_grad()``, never
    ``optimizer.zero_grad()`` (whose default sets the gradients to None)."""

    def __init__(self, model, criterion, optimizer, process_group=None):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        first = self.params[0]
        if isinstance(optimizer, FlatAdam) and optimizer.bucket.numel() == self.numel:
            optimizer.check_views()
            self.bucket = optimizer.bucket         # the optimizer's flat gradient buffer IS the all-reduce bucket
        else:
            self.bucket = torch.zeros(self.numel, dtype=first.dtype, device=first.device)
            off = 0
            for p in self.params:
                n = p.numel()
                p.grad = self.bucket[off:off + n].view_as(p)
                off += n
        if self.world > 1:
            for p in self.params:          # identical start on every rank
                dist.broadcast(p.data, src=0, group=process_group)

    def zero_grad(self):
        self.bucket.zero_()

    def _check_views(self):
        """The gradients must still be the bucket's views (an ``optimizer.zero_grad()`` by the caller would have replaced them)."""
        base = self.bucket.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + off * self.bucket.element_size():
                raise RuntimeError("DataParallelTrainer: a parameter's .grad is no longer a view of the flat bucket "
                                   "(use trainer.zero_grad(), not optimizer.zero_grad())")
            off += p.numel()

    def shard(self, batch):
        """Rank's slice of a global clip batch (B must divide evenly)."""
        if self.world == 1:
            return batch
        rank = dist.get_rank(self.group)
        B = batch[0].shape[0]
        assert B % self.world == 0, "global batch must be divisible by the number of ranks"
        per = B // self.world
        return tuple(t[rank * per:(rank + 1) * per] for t in batch)

    def _allreduce_gradients(self):
        dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
        self.bucket.div_(self.world)

    def graphed(self, example_local_batch, **kw):
        """The data-parallel step captured in a HIP graph (``GraphedTrainStep``) with the flat-bucket all-reduce inside;
        needs the RCCL backend (gloo collectives cannot be captured) and an optimizer created with capturable=True."""
        return GraphedTrainStep(self.model, self.criterion, self.optimizer, example_local_batch,
                                all_reduce=self._allreduce_gradients if self.world > 1 else None,
                                zero_grad=self.zero_grad, **kw)

    def step(self, local_batch, **kw):
        input, flow, target = local_batch
        self._check_views()
        self.zero_grad()
        loss, loss_sum = clip_loss(self.model, self.criterion, input, flow, target, **kw)
        backward(loss)
        if self.world > 1:
            self._allreduce_gradients()
        self.optimizer.step()
        return float(loss_sum.item()) / target.shape[1]


# ---- the driver around the step: epochs, test pass, checkpoints, restore (mainVideoUnshaded.py:344-375, 397-473, 639-726,
# 799-826) -------------------------------------------------------------------------------------------------------------
def evaluate(model, criterion, loader, device, initial_image="zero", upscale=4, upsample="bilinear", disable_temporal=False):
    """``test(epoch)`` of mainVideoUnshaded.py:639-726: the clip recurrence without gradients over the held-out
    batches; returns ``{'total_loss', 'psnr', "('l1', 'mask')", ...}`` averaged over batches x frames, with
    ``psnr = 10 log10(1 / max(1e-10, mse_color))`` per frame (``:693``).  The reference reads every term of every frame
    back with ``.item()``; here everything is accumulated on the device and read once."""
    was_training = model.training
    model.eval()
    lazy_before = getattr(criterion, 'lazy_values', None)
    if lazy_before is not None:
        criterion.lazy_values = True
    feed_inputs = getattr(criterion, 'uses_input', True)
    kw = dict(mode=upsample, **({"align_corners": False} if upsample in ("bilinear", "bicubic") else {}))
    sums, frames = {}, 0

    def add(key, value):
        value = value.detach().double() if torch.is_tensor(value) else torch.tensor(float(value), dtype=torch.float64, device=device)
        sums[key] = value if key not in sums else sums[key] + value
    with torch.no_grad():
        for batch in loader:
            input, flow, target = (t.to(device) for t in batch)
            B, T, Cout, Hh, Wh = target.shape
            previous_output = previous_input = input_high = None
            for j in range(T):
                if j == 0 or disable_temporal:
                    previous_warped = initialImage(input[:, 0], Cout, initial_image, False, upscale)
                    previous_warped_loss = target[:, 0]
                    if feed_inputs:
                        previous_input = F.interpolate(input[:, 0], size=(Hh, Wh), **kw)
                else:
                    previous_warped = VideoTools.warp_upscale(previous_output, flow[:, j - 1], upscale, special_mask=True)
                    previous_warped_loss = previous_warped
                    if feed_inputs:
                        previous_input = VideoTools.warp_upscale(F.interpolate(input[:, j - 1], size=(Hh, Wh), **kw),
                                                                 flow[:, j - 1], upscale, special_mask=True)
                single_input = torch.cat((input[:, j], VideoTools.flatten_high(previous_warped, upscale)), dim=1)
                prediction, _ = model(single_input)
                if feed_inputs:
                    input_high = F.interpolate(input[:, j], size=(Hh, Wh), **kw)
                loss0, values = criterion(target[:, j], prediction, input_high, previous_input, previous_warped_loss)
                add('total_loss', loss0)
                mse = values[('mse', 'color')]
                mse = mse.detach().double() if torch.is_tensor(mse) else torch.tensor(float(mse), dtype=torch.float64, device=device)
                add('psnr', 10.0 * torch.log10(1.0 / torch.clamp(mse, min=1e-10)))
                for key, value in values.items():
                    add(str(key), value)
                previous_output = torch.cat([torch.clamp(prediction[:, 0:1], -1, +1),
                                             ScreenSpaceShading.normalize(prediction[:, 1:4], dim=1),
                                             torch.clamp(prediction[:, 4:5], 0, +1),
                                             torch.clamp(prediction[:, 5:6], 0, +1)], dim=1)
                frames += 1
    if lazy_before is not None:
        criterion.lazy_values = lazy_before
    model.train(was_training)
    if not frames:
        return {}
    keys = list(sums)
    host = torch.stack([sums[k] for k in keys]).cpu().tolist()       # the one read-back
    return {k: v / frames for k, v in zip(keys, host)}


def checkpoint_path(modeldir, epoch):
    return os.path.join(modeldir, "model_epoch_{}.pth".format(epoch))


def save_checkpoint(modeldir, epoch, model, parameters, optimizer, scheduler):
    """``checkpoint(epoch)`` of mainVideoUnshaded.py:799-811: the WHOLE model, optimizer and scheduler objects plus the
    option dict, keyed as ``inference.LoadedModel`` (and the reference's restore) read them."""
    os.makedirs(modeldir, exist_ok=True)
    path = checkpoint_path(modeldir, epoch)
    opt_dict = dict(parameters) if isinstance(parameters, dict) else dict(vars(parameters))      # `opt_dict = vars(opt)`, :167
    state = {'epoch': epoch + 1, 'model': model, 'parameters': opt_dict, 'optimizer': optimizer, 'scheduler': scheduler}
    torch.save(state, path)
    return path


def find_restore_epoch(modeldir, restore_epoch=-1):
    """``--restoreEpoch -1``: the last consecutively numbered ``model_epoch_N.pth`` (mainVideoUnshaded.py:350-358)."""
    if restore_epoch != -1:
        return restore_epoch
    n = 0
    while os.path.exists(checkpoint_path(modeldir, n + 1)):
        n += 1
    return n


def load_checkpoint(modeldir, restore_epoch=-1, device="cpu"):
    """``--restore`` of mainVideoUnshaded.py:344-375 -> (checkpoint dict, start epoch).  The file is read through the
    restricted unpickler of ``inference.loadedmodel``.  As in the reference the loop then starts AT the restored epoch
    number (``startEpoch = restoreEpoch``), i.e. that epoch is run again on top of its own checkpoint."""
    from .inference.loadedmodel import _CheckpointPickle
    epoch = find_restore_epoch(modeldir, restore_epoch)
    path = checkpoint_path(modeldir, epoch)
    if epoch < 1 or not os.path.exists(path):
        raise FileNotFoundError("no checkpoint to restore in %s (looked for %s)" % (modeldir, path))
    return torch.load(path, map_location=device, weights_only=False, pickle_module=_CheckpointPickle), epoch


def fit(model, criterion, train_loader, test_loader, modeldir, n_epochs, parameters, device=None, lr=1e-4, lr_step=500, lr_gamma=0.5,
        restore=False, restore_epoch=-1, graphed=None, flat_adam=False, log=print, **kw):
    """The epoch loop of mainVideoUnshaded.py:819-826 for the non-adversarial recipe:

        for epoch in range(start, n_epochs + 1):  trainNormal(epoch); test(epoch); checkpoint(epoch)

    ``trainNormal`` (:397-473) = ``scheduler.step()`` at the epoch's start, then one optimisation step per batch
    (``train_step``; on the GPU the step of a full-size batch is captured once in a HIP graph, ``GraphedTrainStep``, and
    replayed; a ragged last batch runs eagerly).  ``parameters``: the option dict / Namespace written into every
    checkpoint (``initialImage``, ``upsample``, ...; ``inference.LoadedModel`` reads it).  ``restore``: continue from
    the newest (or ``restore_epoch``) checkpoint of ``modeldir``, taking model, optimizer and scheduler from it.
    Returns (model, history) -- history = one dict per epoch with the mean training loss, the learning rate, the test
    dict of ``evaluate`` and the checkpoint path.  ``kw`` goes to ``clip_loss`` (initial_image, upscale, ...)."""
    if device is None:
        device = next(model.parameters()).device
    on_gpu = str(device).startswith("cuda")
    if graphed is None:
        graphed = on_gpu
    start = 1
    if restore:
        ckpt, start = load_checkpoint(modeldir, restore_epoch, device)
        model, optimizer, scheduler = ckpt['model'].to(device), ckpt['optimizer'], ckpt['scheduler']
        if isinstance(optimizer, FlatAdam):
            optimizer.rebind()                 # parameters and gradients as views of the flat buffers again (no-op if they still are)
        log("Restore training from %s and epoch %d" % (modeldir, start))
    else:
        model = model.to(device)
        optimizer, scheduler = make_optimizer(model, lr, lr_step, lr_gamma, capturable=on_gpu and graphed, tensor_lr=on_gpu and graphed,
                                              flat=flat_adam)
    eval_kw = {k: v for k, v in kw.items() if k in ("initial_image", "upscale", "upsample", "disable_temporal")}
    history = []
    step_fn, step_shape = None, None
    for epoch in range(start, n_epochs + 1):
        lr_now = step_scheduler(optimizer, scheduler)
        model.train()
        total, batches = None, 0
        for batch in train_loader:
            batch = tuple(t.to(device, non_blocking=True) for t in batch)
            shape = tuple(tuple(t.shape) for t in batch)
            if graphed and on_gpu:
                if step_fn is None:
                    step_fn, step_shape = GraphedTrainStep(model, criterion, optimizer, batch, **kw), shape
                if shape == step_shape:
                    loss = step_fn(batch).detach().clone()
                else:
                    loss = torch.tensor(train_step(model, criterion, optimizer, batch, **kw), device=device)
            else:
                loss = torch.tensor(train_step(model, criterion, optimizer, batch, **kw))
            total = loss if total is None else total + loss
            batches += 1
        train_loss = float(total.item()) / max(1, batches) if total is not None else float('nan')
        log("===> Epoch {} Complete: Avg. Loss: {:.4f}, lr {:.3g}".format(epoch, train_loss, lr_now))
        test = evaluate(model, criterion, test_loader, device, **eval_kw) if test_loader is not None else {}
        if test:
            log("===> Avg. PSNR: {:.4f} dB".format(test['psnr']))
        path = save_checkpoint(modeldir, epoch, model, parameters, optimizer, scheduler)
        log("Checkpoint saved to {}".format(path))
        history.append({'epoch': epoch, 'train_loss': train_loss, 'lr': lr_now, 'test': test, 'checkpoint': path})
    return model, history
