"""Statistics harness: run super-resolution models over a set of clips and tabulate PSNR / MS-SSIM / down-sampling consistency per
channel group -- what ``SuperresolutionNetwork/mainPSNR3_AllStats.py`` does.

Restated pieces (reference file:line):

* baseline "models" nearest / bilinear / bicubic (``SimpleUpsample``, ``:71-97``);
* per clip and model the temporal recurrence of ``:302-346``: frame 0 starts from ``initialImage(.., 'zero')``, frame j > 0 from the
  previous prediction warped with the dataset's flow ``flow[j - 1]`` (``warp_upscale(.., special_mask=True)``), the prediction is
  clamped / normalised (mask to [-1, 1], unit normals, depth and AO to [0, 1]) and fed back;
* ``Statistics`` (``:129-299``): shading with and without ambient occlusion (the set-up of ``:104-116``), a border of 15 low-resolution
  pixels cut off, frames whose ground-truth mask covers less than 5 % skipped, masked PSNR (``utils/psnr.py``) of normal / depth / AO /
  colour, MS-SSIM (``utils/ssim.py``) of the same groups after the prediction was blended with the ground truth outside the mask, the
  L2 distance between the low-resolution input and the down-sampled prediction (normal, colour), L1-error histograms with 200 bins;
* output: one ``Stats_<dataset>_<model>.txt`` per model -- a header and ONE ROW PER CLIP with the 14 tab-separated columns of
  ``:160-163,270-281`` -- and one ``Histogram_<dataset>_<model>.txt`` (``:283-299``).

Metric precision.  The reference evaluates PSNR and MS-SSIM in fp32 on whatever device it runs on.  MS-SSIM forms local variances
as E[x^2] - E[x]^2 under an 11-tap window: on a nearly constant channel (depth, AO) that is a cancellation against C2 = 9e-4, and two
fp32 convolution implementations -- torch's CPU kernel and its device kernel -- disagree by up to 3e-4 in the MS-SSIM of IDENTICAL
images (measured on the bilinear baseline, where no kernel of this package runs).  ``metric_dtype`` (default float64) is the
precision the metrics are evaluated in: in fp64 the table depends on the predictions only, so the HIP run and the CPU run of the
same model agree to 1e-5 (``tests/test_stats_gpu.py``); ``metric_dtype=torch.float32`` is the reference's arithmetic.

Added here: every per-clip quantity also goes into a ``utils.MeanVariance`` accumulator per model (``utils/mv.py``), returned by
``run_statistics`` and written as ``Summary_<dataset>.txt`` (mean and variance over the clips) -- the reference leaves that
aggregation to a spreadsheet.

On a CUDA device the networks run on the HIP kernels (``models.EnhanceNet.forward`` -> ``ops.conv3x3`` ...), the warp is the module
path's (bit-identical to the frame pipeline's fused kernel).  This is an OFFLINE renderer in the sense of INTEGRATION.md section 4:
the last frame of every clip is followed by ``ops.guards_flush``.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import models
from .inference.loadedmodel import guarded_forward
from .utils import MSSSIM, PSNR, MeanVariance, ScreenSpaceShading, initialImage

UPSCALING = 4
BORDER = 15
MIN_FILLING = 0.05
NUM_BINS = 200
COLUMNS = ("PSNR-normal", "PSNR-depth", "PSNR-ao", "PSNR-color-noAO", "PSNR-color-withAO",
           "SSIM-normal", "SSIM-depth", "SSIM-ao", "SSIM-color-noAO", "SSIM-color-withAO",
           "L2-ds-normal-mean", "L2-ds-normal-max", "L2-ds-color-noAO-mean", "L2-ds-color-noAO-max")


class SimpleUpsample(nn.Module):
    """The interpolation baselines (``mainPSNR3_AllStats.py:71-97``): the five input channels resized, AO = 1."""

    def __init__(self, upscale_factor, upsample):
        super().__init__()
        self.upscale_factor, self.upsample = upscale_factor, upsample
        self.input_channels, self.output_channels = 5, 6

    def forward(self, inputs):
        inputs = inputs[:, 0:self.input_channels]
        size = [inputs.shape[2] * self.upscale_factor, inputs.shape[3] * self.upscale_factor]
        kw = {} if self.upsample == "nearest" else {"align_corners": False}
        resized = F.interpolate(inputs, size=size, mode=self.upsample, **kw)
        ones = torch.ones(resized.shape[0], self.output_channels - self.input_channels, resized.shape[2], resized.shape[3],
                          dtype=resized.dtype, device=resized.device)
        return torch.cat([resized, ones], dim=1), None


def default_shading(device):
    """``mainPSNR3_AllStats.py:104-116``."""
    sh = ScreenSpaceShading(device)
    sh.fov(30)
    sh.ambient_light_color(np.array([0.1, 0.1, 0.1]))
    sh.diffuse_light_color(np.array([1.0, 1.0, 1.0]))
    sh.specular_light_color(np.array([0.0, 0.0, 0.0]))
    sh.specular_exponent(16)
    sh.light_direction(np.array([0.1, 0.1, 1.0]))
    sh.material_color(np.array([1.0, 0.3, 0.0]))
    sh.ambient_occlusion(1.0)
    sh.inverse_ao = False
    return sh


class Statistics:
    """Accumulators of one model (``mainPSNR3_AllStats.py:129-299``).  ``add_timestep_sample`` per frame, ``write_sample`` per clip."""

    def __init__(self, device, shading=None, upscaling=UPSCALING, border=BORDER, min_filling=MIN_FILLING, ao_strength=1.0,
                 metric_dtype=torch.float64):
        self.device = device
        self.metric_dtype = metric_dtype
        self.shading = shading if shading is not None else default_shading(device)
        self.upscaling, self.border, self.min_filling, self.ao_strength = upscaling, border, min_filling, ao_strength
        self.ssim = MSSSIM().to(device)
        self.psnr = PSNR().to(device)
        self.histograms = {k: np.zeros(NUM_BINS, dtype=np.float64) for k in ("mask", "normal", "depth", "ao", "color_withAO", "color_noAO")}
        self.histogram_counter = 0
        self.clips = {c: MeanVariance() for c in COLUMNS}          # over the clips written so far
        self.reset()

    def reset(self):
        self.n = 0
        self.sums = dict.fromkeys(COLUMNS, 0.0)

    @staticmethod
    def write_header(file):
        file.write("\t".join(COLUMNS) + "\n")

    def _downsample(self, t):
        # nn.Upsample(scale_factor=1/UPSCALING, mode='bilinear') of :133-134
        return F.interpolate(t, scale_factor=1.0 / self.upscaling, mode='bilinear', align_corners=False)

    def add_timestep_sample(self, pred_mnda, gt_mnda, input_mnda):
        """pred / gt: [1, 6, H, W] mask, normal, depth, AO at the high resolution; input: [1, 5, h, w] the low-resolution frame."""
        sh = self.shading
        sh.ambient_occlusion(self.ao_strength)
        pred_c_ao, gt_c_ao = sh(pred_mnda), sh(gt_mnda)
        sh.ambient_occlusion(0.0)
        pred_c, gt_c, in_c = sh(pred_mnda), sh(gt_mnda), sh(input_mnda)
        sh.ambient_occlusion(self.ao_strength)
        b, b2 = self.border, self.border * self.upscaling
        cut = (lambda t, k: t[:, :, k:-k, k:-k]) if b > 0 else (lambda t, k: t)
        pred_mnda, pred_c_ao, pred_c = cut(pred_mnda, b2), cut(pred_c_ao, b2), cut(pred_c, b2)
        gt_mnda, gt_c_ao, gt_c = cut(gt_mnda, b2), cut(gt_c_ao, b2), cut(gt_c, b2)
        input_mnda, in_c = cut(input_mnda, b), cut(in_c, b)
        md = self.metric_dtype                                      # shading above runs in the tensors' own precision; the METRICS in `md`
        pred_mnda, pred_c_ao, pred_c, gt_mnda, gt_c_ao, gt_c, input_mnda, in_c = (
            t.to(md) for t in (pred_mnda, pred_c_ao, pred_c, gt_mnda, gt_c_ao, gt_c, input_mnda, in_c))
        mask = gt_mnda[:, 0:1] * 0.5 + 0.5
        _, _, H, W = mask.shape
        if torch.sum(mask).item() / (H * W) < self.min_filling:
            return False                                            # too few filled pixels (:208-211)
        self.n += 1
        s = self.sums
        s["PSNR-normal"] += self.psnr(pred_mnda[:, 1:4], gt_mnda[:, 1:4], mask=mask).item()
        s["PSNR-depth"] += self.psnr(pred_mnda[:, 4:5], gt_mnda[:, 4:5], mask=mask).item()
        s["PSNR-ao"] += self.psnr(pred_mnda[:, 5:6], gt_mnda[:, 5:6], mask=mask).item()
        s["PSNR-color-withAO"] += self.psnr(pred_c_ao, gt_c_ao, mask=mask).item()
        s["PSNR-color-noAO"] += self.psnr(pred_c, gt_c, mask=mask).item()
        pred_mnda = gt_mnda + mask * (pred_mnda - gt_mnda)          # SSIM sees the ground truth outside the mask (:223)
        s["SSIM-normal"] += self.ssim(pred_mnda[:, 1:4], gt_mnda[:, 1:4]).item()
        s["SSIM-depth"] += self.ssim(pred_mnda[:, 4:5], gt_mnda[:, 4:5]).item()
        s["SSIM-ao"] += self.ssim(pred_mnda[:, 5:6], gt_mnda[:, 5:6]).item()
        s["SSIM-color-withAO"] += self.ssim(pred_c_ao, gt_c_ao).item()
        s["SSIM-color-noAO"] += self.ssim(pred_c, gt_c).item()
        ds_normal = (input_mnda[:, 1:4] - ScreenSpaceShading.normalize(self._downsample(pred_mnda[:, 1:4]), dim=1)) ** 2
        ds_color = (in_c - self._downsample(pred_c)) ** 2
        s["L2-ds-normal-mean"] += torch.mean(ds_normal).item()
        s["L2-ds-normal-max"] = max(s["L2-ds-normal-max"], torch.max(ds_normal).item())
        s["L2-ds-color-noAO-mean"] += torch.mean(ds_color).item()
        s["L2-ds-color-noAO-max"] = max(s["L2-ds-color-noAO-max"], torch.max(ds_color).item())
        self.histogram_counter += 1
        for key, diff in (("mask", (gt_mnda[0, 0] - pred_mnda[0, 0]).abs()),
                          ("normal", (gt_mnda[0, 1:4] - pred_mnda[0, 1:4]).abs().sum(dim=0) / 6),
                          ("depth", (gt_mnda[0, 4] - pred_mnda[0, 4]).abs()), ("ao", (gt_mnda[0, 5] - pred_mnda[0, 5]).abs()),
                          ("color_withAO", (gt_c_ao[0, 0] - pred_c_ao[0, 0]).abs()), ("color_noAO", (gt_c[0, 0] - pred_c[0, 0]).abs())):
            h, _ = np.histogram(diff.detach().cpu().numpy(), bins=NUM_BINS, range=(0, 1), density=True)
            self.histograms[key] += (h / NUM_BINS - self.histograms[key]) / self.histogram_counter
        return True

    def sample_row(self):
        n = max(1, self.n)
        return [self.sums[c] if c.endswith("-max") else self.sums[c] / n for c in COLUMNS]

    def write_sample(self, file):
        """All frames of a clip were added: one row (``:270-281``), fold it into the per-model MeanVariance accumulators, reset."""
        row = self.sample_row()
        file.write("\t".join(("%.6f" % v) if k < 10 else ("%e" % v) for k, v in enumerate(row)) + "\n")
        file.flush()
        if self.n > 0:
            for c, v in zip(COLUMNS, row):
                self.clips[c].append(v)
        self.reset()
        return row

    def write_histogram(self, file):
        file.write("BinStart\tBinEnd\tL2ErrorMask\tCosineErrorNormal\tL2ErrorDepth\tL2ErrorAO\tL2ErrorColorWithAO\tL2ErrorColorNoAO\n")
        hs = self.histograms
        for i in range(NUM_BINS):
            file.write("%7.5f\t%7.5f\t%e\t%e\t%e\t%e\t%e\t%e\n" % (i / NUM_BINS, (i + 1) / NUM_BINS, hs["mask"][i], hs["normal"][i],
                                                                  hs["depth"][i], hs["ao"][i], hs["color_withAO"][i], hs["color_noAO"][i]))


def clip_files(folder):
    """(low, high, flow) paths of the consecutively numbered clips of ``folder`` (``:312-318``)."""
    out = []
    for i in range(10000):
        low = os.path.join(folder, "low_%05d.npy" % i)
        if not os.path.isfile(low):
            break
        out.append((low, os.path.join(folder, "high_%05d.npy" % i), os.path.join(folder, "flow_%05d.npy" % i)))
    return out


def load_models(specs, device, upscaling=UPSCALING):
    """specs: [{'name': .., 'path': checkpoint or None (name = nearest | bilinear | bicubic) or 'model': an nn.Module}]
    -> [(name, module)] (``:99-105``: ``inference.LoadedModel(path).model``)."""
    from .inference import LoadedModel
    out = []
    for m in specs:
        if m.get("model") is not None:
            net = m["model"].to(device).eval()
        elif m.get("path"):
            net = LoadedModel(m["path"], device, upscaling).model
        else:
            net = SimpleUpsample(upscaling, m["name"]).to(device)
        out.append((m["name"], net))
    return out


def run_clip(net, low, high, flow, stats, upscaling=UPSCALING):
    """One clip through one model with the recurrence of ``:326-371``; returns the clip's row."""
    nf = low.shape[0]
    previous_output = None
    for j in range(nf):
        if j == 0:
            previous_warped = initialImage(low[0:1], 6, 'zero', False, upscaling)
        else:
            previous_warped = models.VideoTools.warp_upscale(previous_output, flow[j - 1:j], upscaling, special_mask=True)
        single_input = torch.cat((low[j:j + 1], models.VideoTools.flatten_high(previous_warped, upscaling)), dim=1)
        # (on the device: with the guard contract of LoadedModel.inference around the call -- poll, first-frame range check, publish;
        # run_statistics flushes the last frame's words at the end of the clip)
        prediction = net(single_input)[0] if isinstance(net, SimpleUpsample) else guarded_forward(net, single_input)
        prediction = torch.cat([torch.clamp(prediction[:, 0:1], -1, +1), ScreenSpaceShading.normalize(prediction[:, 1:4], dim=1),
                                torch.clamp(prediction[:, 4:6], 0, +1)], dim=1)
        stats.add_timestep_sample(prediction, high[j:j + 1], low[j:j + 1])
        previous_output = prediction
    return stats


def run_statistics(datasets, model_specs, output_folder, device="cuda", upscaling=UPSCALING, border=BORDER, min_filling=MIN_FILLING,
                   log=print, metric_dtype=torch.float64):
    """``datasets``: [(name, [folders])] (``:29-41``); ``model_specs``: see ``load_models``.  Writes ``Stats_<dataset>_<model>.txt``,
    ``Histogram_<dataset>_<model>.txt`` and ``Summary_<dataset>.txt`` into ``output_folder``; returns
    {dataset: {model: {column: (mean, variance, clips)}}}."""
    os.makedirs(output_folder, exist_ok=True)
    nets = load_models(model_specs, device, upscaling)
    is_cuda = str(device).startswith("cuda")
    result = {}
    for dataset_name, folders in datasets:
        log("Compute statistics for", dataset_name)
        files = [open(os.path.join(output_folder, "Stats_%s_%s.txt" % (dataset_name, name)), "w") for name, _ in nets]
        stats = [Statistics(device, upscaling=upscaling, border=border, min_filling=min_filling, metric_dtype=metric_dtype) for _ in nets]
        try:
            for f in files:
                Statistics.write_header(f)
            with torch.no_grad():
                for folder in folders:
                    for p_low, p_high, p_flow in clip_files(folder):
                        low, high, flow = (torch.from_numpy(np.load(p)).to(device) for p in (p_low, p_high, p_flow))
                        for (name, net), st, f in zip(nets, stats, files):
                            st.reset()
                            if is_cuda:
                                from . import ops
                                # every (model, clip) starts like a freshly loaded model: guard words handed out anew, the clip's FIRST frame gets
                                # the synchronous range check (LoadedModel._setup does the same for a checkpoint) -- several models take turns here
                                ops.range_reset()
                            run_clip(net, low, high, flow, st, upscaling)
                            if is_cuda:
                                from . import ops
                                ops.guards_flush(device)           # the clip's last frame is looked at too
                            st.write_sample(f)
            for (name, _), st in zip(nets, stats):
                with open(os.path.join(output_folder, "Histogram_%s_%s.txt" % (dataset_name, name)), "w") as hf:
                    st.write_histogram(hf)
        finally:
            for f in files:
                f.close()
        summary = {name: {c: (st.clips[c].mean(), st.clips[c].var(), st.clips[c].count()) for c in COLUMNS} for (name, _), st in zip(nets, stats)}
        with open(os.path.join(output_folder, "Summary_%s.txt" % dataset_name), "w") as sf:
            sf.write("model\tclips\t" + "\t".join("%s-mean\t%s-var" % (c, c) for c in COLUMNS) + "\n")
            for name, cols in summary.items():
                sf.write("%s\t%d\t" % (name, cols[COLUMNS[0]][2]) + "\t".join("%.6f\t%e" % (cols[c][0], cols[c][1]) for c in COLUMNS) + "\n")
        result[dataset_name] = summary
    return result


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="PSNR / MS-SSIM statistics of super-resolution models over clip folders (mainPSNR3_AllStats.py)")
    ap.add_argument("--dataset", action="append", required=True, help="name=folder[,folder...] (repeatable)")
    ap.add_argument("--model", action="append", default=[], help="name=checkpoint.pth (repeatable); nearest / bilinear / bicubic are always included")
    ap.add_argument("--output", default="results")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    datasets = [(d.split("=", 1)[0], d.split("=", 1)[1].split(",")) for d in args.dataset]
    specs = [{"name": n, "path": None} for n in ("nearest", "bilinear", "bicubic")]
    specs += [{"name": m.split("=", 1)[0], "path": m.split("=", 1)[1]} for m in args.model]
    res = run_statistics(datasets, specs, args.output, device=args.device)
    for ds, per_model in res.items():
        for name, cols in per_model.items():
            print("%s / %s: PSNR-normal %.3f dB, SSIM-normal %.5f over %d clips" % (ds, name, cols["PSNR-normal"][0], cols["SSIM-normal"][0], cols["PSNR-normal"][2]))


if __name__ == "__main__":
    main()
