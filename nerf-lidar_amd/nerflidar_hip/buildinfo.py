"""Identity of the kernel sources a binary / a profile belongs to.

`kernel_source_sha()` hashes everything `make -C nerf-lidar_amd` compiles (csrc/*, the public header, the Makefile).  The Makefile
compiles that value INTO libnerflidar_hip.so; `binary_sha()` reads it back (`nlr_build_sha()`).  bench.py stamps its JSON line
with the BINARY's value and only quotes PMC traffic from a profile that recorded the same one (scripts/pmc_traffic.sh writes it),
so neither a stale profile nor a stale library can pass for a measurement of the benched code; `stale()` names the mismatch."""
from __future__ import annotations

import glob
import hashlib
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
    files = sorted(glob.glob(os.path.join(_PKG, "csrc", "*")))
    files += [os.path.join(_PKG, "Makefile"), os.path.join(os.path.dirname(_PKG), "include", "nerflidar_hip.h")]
    return [f for f in files if os.path.isfile(f)]


def kernel_source_sha(flags_file=None) -> str:
    """Hash of the sources AND of the build's flag stamp (build/flags.txt: compiler, version, architecture, flags, EXTRA - written by the
    Makefile), so that a library built with other flags (e.g. EXTRA=-DNLR_STAMPS) does not carry the benchmark binary's identity."""
    h = hashlib.sha256()
    for f in kernel_source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    flags_file = flags_file or os.path.join(_PKG, "build", "flags.txt")
    if os.path.isfile(flags_file):
        h.update(b"flags\0" + open(flags_file, "rb").read())
    return h.hexdigest()


def binary_sha() -> str:
    """The source hash compiled into the loaded libnerflidar_hip.so."""
    from . import _lib
    return _lib.lib().nlr_build_sha().decode()


def stale():
    """None when the loaded library was built from the sources lying next to it, else a sentence naming the two hashes."""
    b, s = binary_sha(), kernel_source_sha()
    if b == s:
        return None
    return f"libnerflidar_hip.so was built from kernel source {b[:12]}, the sources in the tree are {s[:12]}: the library is stale"


if __name__ == "__main__":
    import sys
    print(kernel_source_sha(sys.argv[1] if len(sys.argv) > 1 else None))
