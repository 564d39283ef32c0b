"""Scan the library's gfx950 assembly for the store-data hazard the compiler does not cover: a
16-byte (or 12-byte) buffer store with its scalar offset in a REGISTER, followed directly by an
instruction that writes one of the store's data registers (the store fetches its data after it
issues; the last lanes then store the new value).  buf_store4_soff in csrc/common.h is the cure.

    python scripts/scan_store_hazard.py            # compiles csrc/erfnet_hip.hip to assembly (~1 min)
    python scripts/scan_store_hazard.py file.s     # scans an existing listing
Exit status 1 when the pattern is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def listing():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    out = os.path.join(tempfile.mkdtemp(), "erfnet_hip.s")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",      # (the library's flags)
                           "--cuda-device-only", "-S", os.path.join(ROOT, "erfnet_pytorch_amd", "csrc", "erfnet_hip.hip"),
                           "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    lines = listing().split("\n")
    store = re.compile(r"^\s+buffer_store_dwordx[34] v\[(\d+):(\d+)\], (\S+), s\[\d+:\d+\], (\S+)")
    kern, hits, stores = None, [], 0
    for n, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        m = store.match(l)
        if not m or not m.group(4).startswith("s"):     # an immediate offset: the compiler places the wait state
            continue
        stores += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        k = n + 1
        while k < len(lines) and (not lines[k].startswith("\t") or lines[k].strip().startswith((";", "."))):
            k += 1
        nxt = lines[k].strip()
        dm = re.match(r"^v_\w+ v(\d+)\b|^v_\w+ v\[(\d+):(\d+)\]", nxt)
        if dm:
            a = int(dm.group(1) or dm.group(2))
            b = int(dm.group(1) or dm.group(3))
            if not (b < lo or a > hi):
                hits.append((kern, n + 1, l.strip(), nxt))
    print(f"{stores} wide buffer stores with a register offset, {len(hits)} followed by a write of their data")
    for h in hits:
        print("  ", *h)
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
