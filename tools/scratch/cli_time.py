import os, sys, subprocess, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lfbm5d_amd import synth
from PIL import Image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
tmp = tempfile.mkdtemp(dir="/tmp")
lf = synth.make_lf(n, n, 512, 512)
src = os.path.join(tmp, "src"); os.makedirs(src)
t0 = time.time()
for s in range(n):
    for t in range(n):
        Image.fromarray(np.ascontiguousarray(lf[s * n + t].reshape(3, 512, 512).transpose(1, 2, 0))).save(f"{src}/SAI_{s + 1:02d}_{t + 1:02d}.png")
print(f"wrote {n*n} PNGs in {time.time()-t0:.1f} s")
for d in ("noisy", "basic", "denoised", "diff"):
    os.makedirs(os.path.join(tmp, d))
args = [os.path.join(ROOT, "lfbm5d_amd", "LFBM5Ddenoising"), src, "SAI", "_", str(n), str(n), "1", "1", "1", "1", "row", "25", "2.7", f"{tmp}/noisy", f"{tmp}/basic",
        f"{tmp}/denoised", f"{tmp}/diff", "8", "18", "6", "16", "4", "id", "sadct", "haar", "0", "16", "18", "6", "8", "4",
        "dct", "sadct", "haar", "0", "opp", "0", f"{tmp}/m.txt"]
t0 = time.time()
out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, LFBM5D_SEED="1") if os.environ.get("CLI_SEEDED", "1") == "1" else dict(os.environ))
print(f"CLI wall {time.time()-t0:.2f} s rc {out.returncode}")
for l in out.stdout.split("\n"):
    if "elapsed" in l or "time" in l.lower() or "done in" in l: print("  ", l.strip()[:120])
