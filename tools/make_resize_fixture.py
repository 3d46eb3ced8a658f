"""Build-container tool: fixture for the device Resize + CenterCrop.  PIL is absent, so the expected
outputs come from torch: F.interpolate(mode="bicubic", antialias=True) on the float image, rne, clamp, then the centre
crop with torchvision's offsets.  Output: tests/golden/resize.npz (inputs + expected uint8 crops)."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)
rng = np.random.default_rng(7)
out = {}
cases = [(90, 120, 64, 56), (150, 100, 64, 56), (40, 60, 64, 56), (131, 97, 80, 70)]      # (H0, W0, size, crop); 3rd upsamples
for ci, (H0, W0, size, crop) in enumerate(cases):
    # smooth-ish content plus noise (natural images are not white noise; ties at .5 are rarer on smooth data)
    yy, xx = np.mgrid[0:H0, 0:W0]
    base = 127 + 90 * np.sin(yy / 9.0)[..., None] * np.cos(xx / 7.0)[..., None] * np.array([1.0, 0.7, -0.8])
    img = np.clip(base + rng.normal(0, 25, (H0, W0, 3)), 0, 255).astype(np.uint8)[None]
    Hr, Wr = (size, int(size * W0 / H0)) if H0 <= W0 else (int(size * H0 / W0), size)
    r = F.interpolate(torch.from_numpy(img).permute(0, 3, 1, 2).float(), size=(Hr, Wr), mode="bicubic", antialias=True,
                      align_corners=False)
    r = r.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    top, left = int(round((Hr - crop) / 2.0)), int(round((Wr - crop) / 2.0))
    out[f"in/{ci}"] = img
    out[f"cfg/{ci}"] = np.array([size, crop], np.int32)
    out[f"out/{ci}"] = r[:, top:top + crop, left:left + crop]
out["n"] = len(cases)
np.savez_compressed(os.path.join(os.path.dirname(HERE), "tests", "golden", "resize.npz"), **out)
print("resize.npz", os.path.getsize(os.path.join(os.path.dirname(HERE), "tests", "golden", "resize.npz")))
