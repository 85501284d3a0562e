#!/usr/bin/env python3
"""Does a process that created CU-masked streams (vtd_stream_create_masked) exit cleanly?  Variants: destroy / leave alive / torch wrapper alive."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import torch
from vtd_amd import _native
lib = _native.require()
mode = sys.argv[3]
m = (C.c_uint32 * 8)(*([0x0f0f0f0f] * 8))
h = C.c_void_p()
_native.check(lib.vtd_stream_create_masked(m, 8, C.byref(h)), "create")
s = torch.cuda.ExternalStream(h.value)
x = torch.randn(1024, 1024, device="cuda")
with torch.cuda.stream(s):
    y = (x @ x).sum()
torch.cuda.synchronize()
print(mode, float(y))
if mode == "destroy":
    del s
    _native.check(lib.vtd_stream_destroy(h), "destroy")
elif mode == "destroy_keep_wrapper":
    _native.check(lib.vtd_stream_destroy(h), "destroy")
elif mode == "leak":
    pass
elif mode == "leak_del_wrapper":
    del s
elif mode == "os_exit":
    sys.stdout.flush()
    import os
    os._exit(0)
"""
for mode in ("destroy", "destroy_keep_wrapper", "leak", "leak_del_wrapper", "os_exit"):
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "video-text-detection-system_amd"), ROOT, mode], capture_output=True, text=True)
    print(f"{mode}: rc={r.returncode} out={r.stdout.strip()} err={r.stderr.strip()[-200:]}")
