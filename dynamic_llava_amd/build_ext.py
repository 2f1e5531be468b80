"""Builds dynamic_llava_amd/csrc/*.hip into dynamic_llava_amd/libdynllava_hip.so with hipcc for gfx950.

In-tree on purpose: the .so travels to the GPU box with the repo snapshot (a JIT cache would not)."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdynllava_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP vectoriser pairs independent fp32 operations into v_pk_{fma,mul,add}_f32 and broadcasts an operand through
# op_sel.  On gfx950 (ROCm 7.2) the form whose LOW half takes src1's HIGH register (op_sel:[.,1,...]) returns wrong values while a wave of
# ANOTHER kernel executes MFMA on the same SIMD (tools/pkfma_probe.hip: 0 mismatches in 1.6e10 threads alone, ~3 % of threads beside an MFMA
# kernel on a second stream; this is what made two ranks sharing one GPU disagree -- DESIGN.md section 5).  Scalar fp32 VALU code has the
# same rounding, so results do not change; tests/test_host_cpu.py disassembles the library and fails if such an instruction comes back.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("HIPCC_EXTRA", "").split()  # measurement builds (e.g. -DDL_LP_ABLATIONS); part of the stamp, so such a library is never mistaken for the product's


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = _sources()
    deps = srcs + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(os.path.dirname(HERE), "include", "dynllava.h")]
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest(deps)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    keep = {os.path.basename(s_) + ".o" for s_ in srcs} | {"stamp"}
    for f in os.listdir(OBJ):  # objects of sources that no longer exist would otherwise sit there forever (they were never linked)
        if f not in keep:
            os.remove(os.path.join(OBJ, f))

    def cc(src):
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
