"""Turn the reference's only data fixture (testing/sourceLF/SAI_0s_0t.png, 3x3 SAIs, 256x256 RGB
8-bit) into tests/golden/sourceLF_3x3_256_u8.npy, shape [9][3][256][256] uint8, st = s*3 + t
(row-major, s = first file index = aheight index; utilities_LF.cpp:105-146).
Run in the development container only (needs /root/reference and PIL)."""
import os
import numpy as np
from PIL import Image

src = "/root/reference/testing/sourceLF"
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "sourceLF_3x3_256_u8.npy")
lf = np.zeros((9, 3, 256, 256), np.uint8)
for s in range(3):
    for t in range(3):
        im = np.asarray(Image.open(f"{src}/SAI_{s + 1:02d}_{t + 1:02d}.png").convert("RGB"))
        assert im.shape == (256, 256, 3)
        lf[s * 3 + t] = im.transpose(2, 0, 1)
np.save(out, lf)
print("wrote", out, lf.shape, lf.dtype)
