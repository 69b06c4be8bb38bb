"""lfbm5d_amd -- MI355X-native LFBM5D denoising core.

Host-side mirror of the reference's operator interface for the hot path
(`run_bm5d_1st_step` / `run_bm5d_2nd_step`, src/bm5d.h:11-62, and the inner `bm5d_1st_step` /
`bm5d_2nd_step`, src/bm5d_core_processing.h:6-80) over the C-ABI of include/lfbm5d.h.  All compute
runs in liblfbm5d_hip.so (hand-written HIP for gfx950); there is no CPU fallback: importing works
anywhere, creating a context without a GPU raises.
"""
from .core import (Context, Params, Bm3dParams, Stats, run_bm5d_1st_step, run_bm5d_2nd_step, run_bm3d_LF, shard_rows,
                   YUV, YCBCR, OPP, RGB, ID, DCT, SADCT, BIOR, HADAMARD, HAAR, ROWMAJOR, COLMAJOR,
                   TAU, COLOR_SPACE, LfBm5dError, library_path, build_library)

__all__ = ["Context", "Params", "Bm3dParams", "Stats", "run_bm5d_1st_step", "run_bm5d_2nd_step", "run_bm3d_LF", "shard_rows",
           "YUV", "YCBCR", "OPP", "RGB", "ID", "DCT", "SADCT", "BIOR", "HADAMARD", "HAAR",
           "ROWMAJOR", "COLMAJOR", "TAU", "COLOR_SPACE", "LfBm5dError", "library_path",
           "build_library"]
