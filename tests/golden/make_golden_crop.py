#!/usr/bin/env python3
"""Golden vectors of the full-frame helper: the reference's CropParameters (utils/inference_utils.py:278-314) — its integer fields and
its ReflectionPad2d output — for the raw DAVIS frame (260 x 346, 3 encoders) and small odd sizes, plus the reference NETWORK run on a
reflect-padded small frame and cropped back (what full-frame mode computes).  Imports the reference; writes tests/golden/crop.npz."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, load_cfg, model_cfg  # noqa: E402
from recipe import make_item  # noqa: E402


def main():
    mm, *_ = import_reference()
    import utils.inference_utils as iu
    rng = np.random.default_rng(31)
    out = {}
    fields = ["height", "width", "num_encoders", "width_crop_size", "height_crop_size", "padding_top", "padding_bottom", "padding_left",
              "padding_right", "cx", "cy", "ix0", "ix1", "iy0", "iy1"]
    for tag, (w, h, ne) in {"davis": (346, 260, 3), "odd": (13, 10, 3), "odd2": (29, 19, 2), "exact": (48, 32, 3)}.items():
        c = iu.CropParameters(w, h, ne)
        out[tag + ".fields"] = np.array([getattr(c, f) for f in fields], np.int64)
        if tag != "davis":
            x = rng.standard_normal((2, 3, h, w)).astype(np.float32)
            out[tag + ".x"], out[tag + ".padded"] = x, c.pad(torch.from_numpy(x)).numpy()
    out["field_names"] = np.array(json.dumps(fields))
    # the reference network on a reflect-padded 26 x 35 frame (-> 32 x 40), predictions cropped back with the reference's window
    ramnet = load_cfg("train_e2depth_si_grad_loss_statenet_ergb.json")
    cfg = model_cfg(ramnet, every_x_rgb_frame=2, loss_composition=["image", "events1"])
    torch.manual_seed(0)
    m = mm.ERGB2DepthRecurrent(cfg)
    m.gpu = torch.device("cpu")
    m.eval()
    H, W = 26, 35
    c = iu.CropParameters(W, H, 3)
    items = [make_item(rng, 1, H, W, 2, 5, 1, False, 0.0) for _ in range(2)]
    prev, lstm = None, {"events0": None, "events1": None, "depth0": None, "depth1": None, "image": None}
    with torch.no_grad():
        for l, item in enumerate(items):
            padded = {k: c.pad(v) for k, v in item.items()}
            preds, supers, lstm = m(padded, prev, lstm)
            prev = supers["image"]
            for k, v in item.items():
                out["net.item%d.%s" % (l, k)] = v.numpy()
            for k, v in preds.items():
                out["net.pred%d.%s" % (l, k)] = v[:, :, c.iy0:c.iy1, c.ix0:c.ix1].numpy()
    out["net.config"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, "crop.npz"), **out)
    print("crop.npz", len(out))


if __name__ == "__main__":
    main()
