#!/usr/bin/env python3
"""A/B of libhcir builds on the device PNG decoder, interleaved rounds in ONE process: python3 tools/ab_png.py tag=path ...
(tag 'base=' = the in-tree library).  Batches of 256, 600, 880 and 1760 of the bench's hair-like files, window 224."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")]
import torch
from hcir import _lib, png
from bench import png_hair_files


def load(path):
    l = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
    return l


def main():
    libs = []
    for spec in sys.argv[1:]:
        tag, _, path = spec.partition("=")
        libs.append((tag, load(path or _lib.LIB_PATH)))
    files = png_hair_files()
    st = torch.cuda.current_stream().cuda_stream
    for batch in (256, 600, 880, 1760):
        fl = [files[i % len(files)] for i in range(batch)]
        sb = png.stage_batch(fl).to("cuda")
        hdrs = sb._host_headers.data_ptr()      # the launcher reads the headers on the host
        out = torch.empty((batch, 224, 224, 3), dtype=torch.uint8, device="cuda")
        status = torch.empty(batch, dtype=torch.int32, device="cuda")
        wsb = max(L.hcir_png_workspace_bytes(hdrs, batch, 224, 224) for _, L in libs)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        call = lambda L: L.hcir_png_decode_window_u8(sb.blob.data_ptr(), hdrs, batch, 224, 224, out.data_ptr(), status.data_ptr(),
                                                     ws.data_ptr(), wsb, st)
        ref = None
        for t, L in libs:
            assert call(L) == 0
            torch.cuda.synchronize()
            assert int(status.abs().sum()) == 0
            ref = out.clone() if ref is None else ref
            assert torch.equal(out, ref), f"{t}: windows differ"
        times = {t: [] for t, _ in libs}
        for r in range(5):
            for t, L in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    call(L)
                e1.record()
                torch.cuda.synchronize()
                times[t].append(e0.elapsed_time(e1) / 3)
        print(f"batch {batch}: " + "   ".join(f"{t} {statistics.median(v):.2f} ms ({batch / statistics.median(v):.1f} k img/s)" for t, v in times.items()),
              flush=True)


if __name__ == "__main__":
    main()
