"""Build dbgvar/lib_dbg.so: the library with run-time ablation switches in igemm_f32_kernel
(ERFNET_IGEMM_DBG bits: 1 = no output stores, 2 = no global loads in the K loop, 4 = no LDS writes,
8 = no MFMAs, 16 = no epilogue).  The switches are patched into a copy of the sources, never into
the tree; scripts/dbg_sweep.sh runs the sweep on the GPU box with ERFNET_HIP_SO pointing at it."""
import os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TMP = "/tmp/erfnet_dbgsrc"


def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        if old not in s:
            sys.exit(f"{path}: pattern not found:\n{old}")
        s = s.replace(old, new, 1)
    open(path, "w").write(s)


shutil.rmtree(TMP, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "erfnet_pytorch_amd", "csrc"), os.path.join(TMP, "csrc"))
os.makedirs(os.path.join(TMP, "include"))
shutil.copy(os.path.join(ROOT, "include", "erfnet_hip.h"), os.path.join(TMP, "include"))
patch(os.path.join(TMP, "csrc", "common.h"),
      [("    int vec_ok;        // in_C, in_coff, K all multiples of 4 and `in` 16-byte aligned\n};",
        "    int vec_ok;\n    int dbg;\n};")])
patch(os.path.join(TMP, "csrc", "igemm.h"), [
    ("        if (s + 1 < nsteps) store_step(buf ^ 1, R);\n        if (s + 2 < nsteps) load_step(s + 2, R);",
     "        if (s + 1 < nsteps && !(d.dbg & 4)) store_step(buf ^ 1, R);\n"
     "        if (s + 2 < nsteps && !(d.dbg & 2)) load_step(s + 2, R);"),
    ("            if (kc0 + q * 8 >= d.K) break;     // wave-uniform: short last chunk",
     "            if (kc0 + q * 8 >= d.K) break;\n            if (d.dbg & 8) continue;"),
    ("                    buf_store1(r_out, ((opx[q] * d.out_C + d.out_coff + cc[b]) * 4u) | rbad[q] | cbad[b], v);",
     "                    if (!(d.dbg & 1)) buf_store1(r_out, ((opx[q] * d.out_C + d.out_coff + cc[b]) * 4u) | rbad[q] | cbad[b], v);"),
    ("    // Epilogue.  C/D layout of the 32x32 MFMA", "    if (d.dbg & 16) return;\n    // Epilogue.  C/D layout of the 32x32 MFMA"),
])
patch(os.path.join(TMP, "csrc", "erfnet_hip.hip"), [
    ('#include "../../include/erfnet_hip.h"', '#include "../include/erfnet_hip.h"'),
    ("    const TileCfg c = kCfg[d.cfg];\n    dim3 grid(",
     "    { const char *e = getenv(\"ERFNET_IGEMM_DBG\"); const_cast<IgemmDesc &>(d).dbg = e ? atoi(e) : 0; }\n"
     "    const TileCfg c = kCfg[d.cfg];\n    dim3 grid("),
])
out = os.path.join(ROOT, "dbgvar")
os.makedirs(out, exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-o", os.path.join(out, "lib_dbg.so"), os.path.join(TMP, "csrc", "erfnet_hip.hip")],
                      stderr=subprocess.DEVNULL)
print("built", os.path.join(out, "lib_dbg.so"))
