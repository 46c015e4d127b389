"""Training-clip dataset of the unshaded pipeline (data contract of SURVEY.md S11).

Restates the array branch of ``SuperresolutionNetwork/datasetVideo.py`` (``collect_samples_clouds_video``
``:84-309`` with ``load_arrays``; ``DatasetFromSamples`` ``:311-366``):

* clips are triples ``high_%05d.npy [T,6,4H,4W]``, ``low_%05d.npy [T,5,H,W]`` (mask in [-1,1], normal,
  depth), ``flow_%05d.npy [T,2,H,W]`` (hole-filled), consecutively numbered in a folder, or in the
  sub-folders listed line by line in a text file;
* a sample is a random 32^2 low-res crop (128^2 high-res) whose first AND last frame are at least
  50 % covered (``:267-283``: coverage = pixels where the sum of the first three low-res channels > 0);
* samples are sorted by clip index and the LAST ``test_fraction`` of them form the test set
  (``:300-301,328-334``) so that train and test never share a clip region ordering;
* optional flip augmentation with the sign fix-ups of ``:31-78`` (off by default as in the reference).
"""
import os
import random
from collections import namedtuple

import numpy as np
import torch
from torch.utils import data

CROP = 32
MAX_AUGMENTATION_MODE = 4
Sample = namedtuple("Sample", "index crop_low crop_high augmentation")
DatasetData = namedtuple("DatasetData", "samples images_high images_low flow_low input_channels output_channels crop_size num_frames")


def _clip_paths(path):
    def names(folder, i):
        return tuple(os.path.join(folder, "%s_%05d.npy" % (m, i)) for m in ("high", "low", "flow"))
    folders = [path]
    if os.path.isfile(path):
        with open(path) as f:
            folders = [os.path.join(os.path.dirname(path), line.strip()) for line in f if line.strip()]
    out = []
    for folder in folders:
        i = 0
        while os.path.exists(names(folder, i)[1]):
            out.append(names(folder, i))
            i += 1
    return out


def data_augmentation(low, high, flow, mode, enabled=False):
    if not enabled:
        return low, high, flow
    flip_x, flip_y = bool(mode & 1), bool(mode & 2)
    axes = tuple(a for a, f in ((2, flip_x), (3, flip_y)) if f)
    if not axes:
        return low, high, flow
    low, high, flow = (np.flip(t, axis=axes).copy() for t in (low, high, flow))
    if low.shape[1] in (7, 8):                      # shaded layouts carry a normal at channels 4,5
        if flip_x: low[:, 4] = -low[:, 4]
        if flip_y: low[:, 5] = -low[:, 5]
    if flip_x: flow[:, 0] = -flow[:, 0]
    if flip_y: flow[:, 1] = -flow[:, 1]
    return low, high, flow


def collect_samples(path, num_samples, upsampling=4, number_of_images=None, seed=None, crop=CROP):
    paths = _clip_paths(path)
    if not paths:
        raise ValueError("No image found")
    if number_of_images:
        paths = paths[:number_of_images]
    high = [np.load(p[0]) for p in paths]
    low = [np.load(p[1]) for p in paths]
    flow = [np.load(p[2]) for p in paths]
    T = low[0].shape[0]
    rng = random.Random(seed)
    fill = 0.5 * crop * crop
    samples = []
    while len(samples) < num_samples:
        idx = rng.randint(0, len(paths) - 1)
        d2, d3 = low[idx].shape[2], low[idx].shape[3]
        x = rng.randint(0, d2 - crop - 1)
        y = rng.randint(0, d3 - crop - 1)
        def covered(t):
            c = low[idx][t, 0:3, x:x + crop, y:y + crop].sum(axis=0) > 0
            return c.sum() >= fill
        if covered(0) and covered(T - 1):
            samples.append(Sample(idx, (x, x + crop, y, y + crop),
                                  (upsampling * x, upsampling * (x + crop), upsampling * y, upsampling * (y + crop)),
                                  rng.randrange(MAX_AUGMENTATION_MODE)))
    samples.sort(key=lambda s: s.index)
    return DatasetData(samples, high, low, flow, 5, high[0].shape[1], crop, T)


class DatasetFromSamples(data.Dataset):
    """``__getitem__`` -> (low [T,5,c,c], flow [T,2,c,c], high [T,6,4c,4c])"""

    def __init__(self, dataset_data, test, test_fraction, augmentation=False):
        self.data = dataset_data
        self.samples = dataset_data.samples
        n = len(self.samples)
        l = int(n * test_fraction)
        self.index_offset, self.num_images = (n - l, l) if test else (0, n - l)
        self.augmentation = augmentation

    def __len__(self):
        return self.num_images

    def __getitem__(self, index):
        s = self.samples[index + self.index_offset]
        cl, ch = s.crop_low, s.crop_high
        low = self.data.images_low[s.index][:, :, cl[0]:cl[1], cl[2]:cl[3]]
        flow = self.data.flow_low[s.index][:, :, cl[0]:cl[1], cl[2]:cl[3]]
        high = self.data.images_high[s.index][:, :, ch[0]:ch[1], ch[2]:ch[3]]
        low, high, flow = data_augmentation(low, high, flow, s.augmentation, self.augmentation)
        return (torch.from_numpy(np.ascontiguousarray(low)), torch.from_numpy(np.ascontiguousarray(flow)),
                torch.from_numpy(np.ascontiguousarray(high)))


def render_clip(renderer, origins, low_res, upscale=4, fov=30.0, isovalue=0.34, ao_samples=0, ao_radius=0.05):
    """Produce one training clip with this package's renderer, in the format above
    (the reference drives GPURenderer.exe for this: ``DataGenerator/DataGeneratorVideo2.py:46-90``)."""
    from .inference.flowfill import fill_flow
    from .volumes import fmt3
    w, h = low_res
    r = renderer
    for c, v in (("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "%.3f" % fov), ("isovalue", "%5.3f" % isovalue),
                 ("aoradius", "%5.3f" % ao_radius)):
        r.send_command(c, v)
    lows, highs, flows = [], [], []
    for origin in origins:
        r.send_command("cameraOrigin", fmt3(origin))
        frames = {}
        for name, (W, H), ao in (("low", (w, h), 0), ("high", (w * upscale, h * upscale), ao_samples)):
            r.send_command("resolution", "%d,%d" % (W, H))
            r.send_command("viewport", "0,0,%d,%d" % (W, H))
            r.send_command("aosamples", "%d" % ao)
            if name == "high":          # keep the flow reference of the low-res stream: re-send the same camera
                pass
            buf = torch.empty((H, W, 12), dtype=torch.float32, device="cuda")
            r.render_direct(buf)
            frames[name] = buf.permute(2, 0, 1)
        lo, hi = frames["low"], frames["high"]
        lows.append(torch.cat((lo[3:4] * 2 - 1, lo[4:8]), 0))
        highs.append(torch.cat((hi[3:4] * 2 - 1, hi[4:8], hi[10:11]), 0))
        flows.append(fill_flow(lo[8:10].unsqueeze(0), lo[3:4].unsqueeze(0) != 0)[0])
    return (torch.stack(highs).cpu().numpy(), torch.stack(lows).cpu().numpy(), torch.stack(flows).cpu().numpy())
