#!/usr/bin/env python3
"""Where an inflate wavefront spends its cycles: run against a library built with -DHCIR_PNG_STAMPS
(tools/build_variant.sh pngstamps "-DHCIR_PNG_STAMPS" png.hip; HCIR_LIB_PATH=tools/_libhcir_pngstamps.so).
Reads the 16 counters per image back from the workspace: s_memtime cycles per phase, symbols, matches, passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")]
import numpy as np
import torch

from hcir import png
from bench_png import hair_like_files


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    files = hair_like_files()
    fl = [files[i % len(files)] for i in range(batch)]
    st = png.stage_batch(fl)
    d = st.to("cuda")
    for _ in range(2):
        png.decode_windows(d, 224)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    png.decode_windows(d, 224)
    e1.record()
    torch.cuda.synchronize()
    print(f"batch {batch}: {e0.elapsed_time(e1):.2f} ms")
    ws = next(iter(png._ws.values()))
    L = png._lib.lib()
    wsb = L.hcir_png_workspace_bytes(st._host_headers.data_ptr(), batch, 224, 224)
    stride = (wsb - 512 - 256 - batch * 132) // batch
    base = (-ws.data_ptr()) % 256
    off = base + ((batch * stride + batch * 4 + 255) & ~255)
    g = ws[off:off + batch * 128].cpu().numpy().view(np.uint64).reshape(batch, 16)
    names = ["other", "header", "lookup", "walk", "-", "-", "glue", "slow"]
    for i in range(min(batch, 8)):
        r = g[i].astype(np.float64)
        tot = r[:8].sum()
        nsym, nmatch, npass, nslow = r[8], r[9], r[10], r[11]
        print(f"img {i}: file {len(fl[i])} B, {nsym:.0f} symbols, {nmatch:.0f} matches, {npass:.0f} passes, "
              f"{tot / 1e6:.1f} Mcycles = {tot / max(nsym, 1):.0f} cyc/symbol")
        print("   " + "  ".join(f"{n} {r[k] / tot * 100:.1f}% ({r[k] / max(nsym, 1):.0f}/sym)" for k, n in enumerate(names)))
        print(f"   copier: waiting {r[13] / max(nsym, 1):.0f}/sym, copying {r[14] / max(nsym, 1):.0f}/sym "
              f"({r[14] / max(nmatch, 1):.0f} cyc/match), write-out {r[15] / max(nsym, 1):.0f}/sym")
        print(f"   lookup {r[2] / max(npass, 1):.0f} cyc/pass, "
              f"{nslow:.0f} serially decoded symbols at {r[7] / max(nslow, 1):.0f} cyc, glue {r[6] / max(npass, 1):.0f} cyc/pass")


if __name__ == "__main__":
    main()
