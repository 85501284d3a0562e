"""Builds vtd_amd/_lib/libvtd_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "vtd_amd", "_lib")
OUT = os.path.join(OUT_DIR, "libvtd_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("VTD_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DVTD_CONV_EXPERIMENT (tools/conv_experiment.sh)
# the post-process geometry replays float32 arithmetic in a fixed order: no fused multiply-add there
PER_FILE = {"postprocess.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
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
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
