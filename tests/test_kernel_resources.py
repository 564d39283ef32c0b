"""Register / scratch budget of the kernels whose speed hangs on it, and a scan of the assembly for a store
hazard the compiler does not cover (CPU tests: hipcc cross-compiles gfx950).

The line kernels keep their weights in registers and run at exactly the occupancy they were designed for;
a change that makes the compiler spill (scratch memory) or lose a wave per SIMD costs 10-15 % of a launch
without failing any parity test.  The device code is compiled once with
`-Rpass-analysis=kernel-resource-usage` (the library's own flags) and the remarks are checked."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernel (demangled prefix) -> (minimum waves per SIMD, maximum scratch bytes per lane)
BUDGET = {
    "void line_x3r_kernel<128, 64>(": (1, 0),
    "void line_x3r_kernel<64, 64>(": (2, 0),
    "void line_x3_kernel<128, 32, false>(": (1, 0),
    "void line2_bf16_kernel<128, 64>(": (2, 0),
    "void line2_bf16_kernel<64, 128>(": (2, 0),
    "void wgrad_line_x3_kernel<128, 2, 2, 2, 2, 1, 1, 96>(": (1, 0),
    "void wgrad_line_x3_kernel<64, 1, 1, 2, 2, 1, 2, 72>(": (2, 0),
    "void wgrad_line_bf16_kernel<128, 1, 2, 4, 2, 1>(": (2, 0),
    "void wgrad_line_bf16_kernel<64, 1, 1, 2, 2, 3>(": (3, 0),
    "void conv16_rows_kernel<4, 4, true>(": (3, 0),
    "void conv16_bf16_rows_kernel<8, 2, true>(": (2, 0),
    "void igemm_x3_kernel<2, 1, 2, 2, 2>(": (2, 0),
    "void up16_fwd_kernel<float, 4, false>(": (3, 0),       # (236 registers, two waves, without the barrier between steps)
    "void up16_fwd_kernel<float, 3, true>(": (3, 0),
    "void up16_fwd_bf16_kernel<4, false>(": (3, 0),         # three blocks per CU
    "void up16_fwd_bf16_kernel<3, false>(": (3, 0),
    "void up16_fwd_bf16_kernel<3, true>(": (3, 0),
    "void outconv_ce_bwd_kernel<5, unsigned short>(": (2, 0),
    "void first_down_fwd_kernel<float>(": (4, 0),
    "void first_down_fwd_kernel<unsigned short>(": (4, 0),
}


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    """(compiler remarks, path of the gfx950 assembly) of the library's device code, compiled once."""
    if not os.path.exists(HIPCC) or shutil.which("c++filt") is None:
        pytest.skip("hipcc / c++filt not available")
    from erfnet_pytorch_amd import _lib
    out = tmp_path_factory.mktemp("res") / "lib.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *_lib.DEVICE_FLAGS, "--cuda-device-only", "-S",
           "-Rpass-analysis=kernel-resource-usage", "-o", str(out),
           os.path.join(ROOT, "erfnet_pytorch_amd", "csrc", "erfnet_hip.hip")]
    text = subprocess.run(cmd, capture_output=True, text=True, timeout=900).stderr
    assert out.exists(), text[-2000:]
    return text, str(out)


@pytest.fixture(scope="module")
def remarks(listing):
    text = listing[0]
    recs = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                      text, re.S)
    assert recs, "no resource remarks in the compiler's output"
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in recs), capture_output=True, text=True).stdout.split("\n")
    return {n: (int(o), int(s)) for n, (_, s, o) in zip(names, recs)}


@pytest.mark.parametrize("kernel", sorted(BUDGET))
def test_kernel_keeps_its_occupancy_and_does_not_spill(remarks, kernel):
    hits = {n: v for n, v in remarks.items() if n.startswith(kernel)}
    assert hits, f"{kernel}...) is not in the library any more: update the table"
    min_waves, max_scratch = BUDGET[kernel]
    for name, (waves, scratch) in hits.items():
        assert waves >= min_waves, f"{name}: {waves} wave(s) per SIMD, designed for {min_waves}"
        assert scratch <= max_scratch, f"{name}: {scratch} bytes of scratch per lane (register spills)"


def test_no_wide_store_is_followed_by_a_write_of_its_data(listing):
    """A 16-byte buffer store whose scalar offset sits in a register fetches its data after it issues and
    the compiler places no wait state behind it: an instruction right behind the store that overwrites a
    data register corrupts the last lanes' values (wrong results that depend on timing, which parity tests
    catch only sometimes).  scripts/scan_store_hazard.py looks for the pattern in the library's assembly."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "scan_store_hazard.py"), listing[1]],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
