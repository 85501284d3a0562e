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

# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs.  Without it hipcc puts the accumulators of the 256-thread kernels into
# AGPRs and shuffles them through v_accvgpr_read / write every K-step (41 copies per 16 MFMAs in conv_igemm<128,64,2,2,2>: the SQ
# counters had that kernel at 7.5 vector instructions per MFMA) and needs 30-50 more registers; the 512-thread kernels already used
# the VGPR form.  Same arithmetic, bit-identical outputs; the 4-wave launches run 2-6 % faster (classed head-entry border tiles 12 %).
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form",
          "-I" + CSRC] + EXTRA
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
    # a change of the compile flags rebuilds everything (the stamp sits beside the objects)
    stamp, flags = os.path.join(OBJ_DIR, "flags.txt"), " ".join(COMMON) + " | " + repr(sorted(PER_FILE.items()))
    if not os.path.exists(stamp) or open(stamp).read() != flags:
        force = True
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
    with open(stamp, "w") as f:
        f.write(flags)
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
