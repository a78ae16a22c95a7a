"""Builds zutis_amd/libzutis_hip.so (all HIP kernels, gfx950) with hipcc.  In-tree, no torch types.

    python -m zutis_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzutis_hip.so")
# float64 bilateral solver: no FMA contraction (bin edges and bistochastisation are bit-compared with NumPy/SciPy)
EXTRA_FLAGS = {"bilateral.hip": ["-ffp-contract=off"]}
# the MFMA kernels live at the register budget of their occupancy: a spill is a 2-3x slowdown, so it is a build error
NO_SCRATCH = {"gemm.hip", "gemm_x3.hip", "attention.hip"}
MAX_SCRATCH = int(os.environ.get("ZH_BUILD_MAX_SCRATCH", "0"))    # bytes per lane tolerated: the MFMA loops must not spill (developer builds may raise it)
# waves per SIMD the design of a kernel relies on (mangled-name substring -> minimum), checked against the compiler's remarks
MIN_OCCUPANCY = {"attn_f16_kernelILi64ELi4ELi1ELi0ELi1E": 3, "attn_f16_kernelILi64ELi4ELi0ELi0ELi1E": 3, "attn_f16_kernelILi64ELi4ELi1ELi1ELi1E": 2,
                 "attn_f16_kernelILi96ELi4ELi1ELi0ELi1E": 2, "attn_f16_kernelILi96ELi4ELi0ELi0ELi1E": 2, "attn_f16_kernelILi96ELi4ELi1ELi1ELi1E": 2,
                 "attn_f16_kernelILi64ELi4ELi1ELi0ELi2E": 2}
SOURCES = ["capi.hip", "gemm.hip", "gemm_x3.hip", "attention.hip", "norm.hip", "resample.hip", "metrics.hip", "instance.hip", "bilateral.hip", "retrieval.hip", "text.hip", "plan.hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_kernel.h"), os.path.join(CSRC, "gemm_skinny.h"),
                        os.path.join(os.path.dirname(HERE), "include", "zutis_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    from . import plan
    plan.generate_dispatch(os.path.join(CSRC, "plan_gen.inc"))      # launch-plan dispatcher, generated from the header
    procs = []
    for src in sources():
        obj = os.path.join(HERE, "_obj", os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        deps = [src, os.path.join(CSRC, "common.h")]
        if os.path.basename(src).startswith("gemm"):
            deps.append(os.path.join(CSRC, "gemm_kernel.h"))
            deps.append(os.path.join(CSRC, "gemm_skinny.h"))
        if os.path.basename(src) in ("plan.hip", "capi.hip"):       # the dispatcher generated from the header / ZH_ABI_VERSION
            deps.append(os.path.join(os.path.dirname(HERE), "include", "zutis_hip.h"))
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps):
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + EXTRA_FLAGS.get(os.path.basename(src), []) + \
              ["-c", src, "-o", obj]
        if os.path.basename(src) in NO_SCRATCH:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, src, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for cmd, src, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(err)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if os.path.basename(src) in NO_SCRATCH:
            spills = [ln for ln in err.splitlines() if "ScratchSize [bytes/lane]:" in ln
                      and int(ln.split("ScratchSize [bytes/lane]:")[1].split()[0]) > MAX_SCRATCH]
            if spills:
                raise RuntimeError(f"{os.path.basename(src)}: a kernel spills to scratch (register budget exceeded): {spills[0].strip()}")
            fn = None
            for ln in err.splitlines():
                if "Function Name:" in ln:
                    fn = ln.split("Function Name:")[1].split()[0]
                elif "Occupancy [waves/SIMD]:" in ln and fn:
                    occ = int(ln.split("Occupancy [waves/SIMD]:")[1].split()[0])
                    for key, need in MIN_OCCUPANCY.items():
                        if key in fn and occ < need:
                            raise RuntimeError(f"{os.path.basename(src)}: {fn} compiles to {occ} waves/SIMD, its design needs {need}")
        else:
            sys.stderr.write(err)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
