"""Identity of the kernel sources a binary / a profile belongs to.

`kernel_source_sha()` hashes everything `make -C nerf-lidar_amd` compiles (csrc/*, the public header, the Makefile).
bench.py stamps it into its JSON line and only quotes PMC traffic from a profile that recorded the same hash
(scripts/pmc_traffic.sh writes it), so a stale profile can never pass for a measurement of the benched code."""
from __future__ import annotations

import glob
import hashlib
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
    files = sorted(glob.glob(os.path.join(_PKG, "csrc", "*")))
    files += [os.path.join(_PKG, "Makefile"), os.path.join(os.path.dirname(_PKG), "include", "nerflidar_hip.h")]
    return [f for f in files if os.path.isfile(f)]


def kernel_source_sha() -> str:
    h = hashlib.sha256()
    for f in kernel_source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_sha())
