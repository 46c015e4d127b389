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


def make_optimizer(model, lr=1e-4, lr_step=500, lr_gamma=0.5, capturable=False):
    opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=capturable)
    sched = torch.optim.lr_scheduler.StepLR(opt, lr_step, lr_gamma)
    return opt, sched


def train_step(model, criterion, optimizer, batch, **kw):
    """One optimisation step on a clip batch (single process)."""
    input, flow, target = batch
    optimizer.zero_grad()
    loss, loss_sum = clip_loss(model, criterion, input, flow, target, **kw)
    backward(loss)
    optimizer.step()
    return float(loss_sum.item()) / target.shape[1]


class GraphedTrainStep:
    """The whole optimisation step (T frames forward with the recurrence, backward through time, Adam) captured ONCE in
    a HIP graph and replayed: at the per-GPU batch of BASELINE config #3 (2 clips) a step is ~1500 small launches and the
    eager loop is bound by Python, not by the GPU.  Needs a GPU, fixed batch shapes, an optimizer created with
    ``capturable=True`` and (as everywhere in this package) no host read-back inside the step; the loss of each step is
    returned as a device tensor.  ``all_reduce`` (optional callable) runs between backward and the optimizer step --
    ``DataParallelTrainer`` passes its flat-bucket RCCL all-reduce, which is captured with the rest."""

    def __init__(self, model, criterion, optimizer, example_batch, all_reduce=None, warmup=3, **kw):
        self.model, self.criterion, self.optimizer, self.kw = model, criterion, optimizer, kw
        self.all_reduce = all_reduce
        self.static = tuple(torch.empty_like(t) for t in example_batch)
        for dst, src in zip(self.static, example_batch):
            dst.copy_(src)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # PyTorch's capture recipe: warm up on a side stream
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def _eager(self):
        input, flow, target = self.static
        self.optimizer.zero_grad(set_to_none=True)
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
        return self.loss


class DataParallelTrainer:
    """Data-parallel ``train_step``: one process per GPU, one flat gradient all-reduce per step."""

    def __init__(self, model, criterion, optimizer, process_group=None):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        first = self.params[0]
        self.bucket = torch.zeros(self.numel, dtype=first.dtype, device=first.device)
        if self.world > 1:
            for p in self.params:          # identical start on every rank
                dist.broadcast(p.data, src=0, group=process_group)

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
        off = 0
        for p in self.params:
            n = p.numel()
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            self.bucket[off:off + n].copy_(g.reshape(-1))
            off += n
        dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
        self.bucket.div_(self.world)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.bucket[off:off + n].view_as(p).clone() if p.grad is None else p.grad.copy_(self.bucket[off:off + n].view_as(p))
            off += n

    def graphed(self, example_local_batch, **kw):
        """The data-parallel step captured in a HIP graph (``GraphedTrainStep``) with the flat-bucket all-reduce inside;
        needs the RCCL backend (gloo collectives cannot be captured) and an optimizer created with capturable=True."""
        return GraphedTrainStep(self.model, self.criterion, self.optimizer, example_local_batch,
                                all_reduce=self._allreduce_gradients if self.world > 1 else None, **kw)

    def step(self, local_batch, **kw):
        input, flow, target = local_batch
        self.optimizer.zero_grad()
        loss, loss_sum = clip_loss(self.model, self.criterion, input, flow, target, **kw)
        backward(loss)
        if self.world > 1:
            self._allreduce_gradients()
        self.optimizer.step()
        return float(loss_sum.item()) / target.shape[1]
