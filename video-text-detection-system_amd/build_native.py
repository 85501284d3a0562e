"""Builds vtd_amd/_lib/libvtd_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "vtd_amd", "_lib")
# Instrumented variants (extra -D flags: tools/conv_experiment.sh) never overwrite the product library: VTD_LIB_VARIANT=<tag>
# builds libvtd_hip_<tag>.so from its own object directory, and vtd_amd/_native.py loads it only when VTD_LIB_VARIANT names it.
VARIANT = os.environ.get("VTD_LIB_VARIANT", "")
EXTRA = os.environ.get("VTD_EXTRA_HIPCC_FLAGS", "").split()
if EXTRA and not VARIANT:
    raise SystemExit("VTD_EXTRA_HIPCC_FLAGS needs VTD_LIB_VARIANT=<tag>: the product library is only ever built with the default flags")
OUT = os.path.join(OUT_DIR, f"libvtd_hip_{VARIANT}.so" if VARIANT else "libvtd_hip.so")
OBJ_DIR = os.path.join(HERE, f"build_{VARIANT}" if VARIANT else "build")

COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-I" + CSRC] + EXTRA
# the post-process geometry replays float32 arithmetic in a fixed order: no fused multiply-add there
PER_FILE = {"postprocess.hip": ["-ffp-contract=off"]}


def sources():
    src = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))
    if "-DVTD_EXPERIMENTAL_CANDIDATES" in EXTRA:   # measured losing kernel candidates: instrumented builds only, never the product library
        src += glob.glob(os.path.join(CSRC, "experimental", "*.hip"))
    return sorted(src)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) +   # .inc: textually included graph code
               glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    newest_hdr = max(os.path.getmtime(h) for h in headers)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr):
            continue
        cmd = ["hipcc", *COMMON, *PER_FILE.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if procs or not os.path.exists(OUT):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT], check=True)
    if not VARIANT:
        try:   # the compute library carries no communication dependency: a host without RCCL still gets libvtd_hip.so
            build_comm(force)
        except (subprocess.CalledProcessError, OSError) as e:
            print(f"build_native: libvtd_comm.so not built ({e}); vtd_gather (include/vtd_comm.h) is unavailable, "
                  "torch.distributed remains the product's collective path", file=sys.stderr)
    return OUT


def build_comm(force=False):
    """libvtd_comm.so (include/vtd_comm.h): the detections all-gather over RCCL, a library of its own (links librccl)."""
    src = os.path.join(HERE, "csrc_comm", "vtd_comm.cpp")
    out = os.path.join(OUT_DIR, "libvtd_comm.so")
    hdr = os.path.join(HERE, "..", "include", "vtd_comm.h")
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        rocm = os.environ.get("ROCM_PATH") or os.environ.get("ROCM_HOME") or "/opt/rocm"
        libdir = os.path.join(rocm, "lib")
        subprocess.run(["hipcc", "-O2", "-fPIC", "-std=c++17", "-Wall", "-shared", src, "-o", out, "-I" + os.path.join(rocm, "include"),
                        "-L" + libdir, "-lrccl", "-Wl,-rpath," + libdir], check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
