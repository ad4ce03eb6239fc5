"""sha256 (first 16 hex digits) over the kernel sources libhcir.so is built from: csrc/*.hip and csrc/*.h in sorted
name order, then include/hcir.h — name and content.  Used by the Makefile (embedded in the binary as hcir_build_id())
and by hcir._lib.source_hash() (the sources on disk); standalone so that `make` needs no torch import."""
import glob
import hashlib
import os
import sys


def source_hash(csrc_dir: str, header: str) -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(csrc_dir, "*.hip")) + glob.glob(os.path.join(csrc_dir, "*.h")))
    files.append(header)
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash(sys.argv[1], sys.argv[2]))
