#!/usr/bin/env python3
"""How does a HIP CU mask (hipExtStreamCreateWithCUMask) map onto the dies of an MI355X?  For a series of masks, launches 4096 short
busy workgroups and counts the distinct CUs (XCC id, shader engine, CU id from the hardware-id register) they ran on, per XCC.
Build: hipcc --offload-arch=gfx950 -shared -fPIC tools/probe/cumask_map.hip -o tools/probe/libcumask_map.so (done here when missing)."""
import ctypes as C
import collections
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "probe", "libcumask_map.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(HERE, "probe", "cumask_map.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.cumask_where.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.c_uint32)]
lib.cumask_where.restype = C.c_int


def cus(bits, blocks=4096):
    m = (C.c_uint32 * 8)()
    for b in bits:
        m[b // 32] |= 1 << (b % 32)
    out = (C.c_uint32 * (2 * blocks))()
    rc = lib.cumask_where(m, 8, blocks, out)
    assert rc == 0, rc
    return {(out[2 * i] & 0xf, (out[2 * i + 1] >> 13) & 7, (out[2 * i + 1] >> 8) & 15) for i in range(blocks)}


def show(name, bits):
    s = cus(bits)
    per = collections.Counter(x for x, _, _ in s)
    print(f"{name}: {len(bits)} bits -> {len(s)} CUs; per XCC {[per.get(x, 0) for x in range(8)]}")
    return s


show("all 256", range(256))
for b in (0, 1, 2, 7, 8, 31, 32, 33, 100, 255):
    s = show(f"bit {b}", [b])
    print("    ", sorted(s)[:8])
for k in (8, 16, 32, 64, 128, 192):
    show(f"bits 0..{k - 1}", range(k))
show("even bits", range(0, 256, 2))
show("bits = 0 mod 4", range(0, 256, 4))
show("bits = 0 mod 8", range(0, 256, 8))
show("bits 0..7 of every 32", [b for b in range(256) if b % 32 < 8])
a = show("low half of every 16 (i % 16 < 8)", [b for b in range(256) if b % 16 < 8])
b = show("high half of every 16", [b for b in range(256) if b % 16 >= 8])
print("disjoint:", not (a & b))
