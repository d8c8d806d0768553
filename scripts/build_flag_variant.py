"""Variant library whose nonbonded.hip is compiled with extra COMPILER flags (scheduling strategies and the like); the other objects are
the product build's.  Selected at run time with TM_AMD_LIB.   python scripts/build_flag_variant.py <tag> <flag> [<flag> ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd.csrc import build as B

tag, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(B.HERE, f"libtimemachine_amd_{tag}.so")
objs = []
for src in B.SOURCES:
    if src != "nonbonded.hip":
        objs.append(os.path.join(B.HERE, os.path.splitext(src)[0] + ".o"))
        continue
    obj = os.path.join(B.HERE, f"nonbonded.{tag}.o")
    r = subprocess.run([B.HIPCC] + B.FLAGS + extra + ["-x", "hip", "-c", os.path.join(B.HERE, src), "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stdout + r.stderr, file=sys.stderr)
        raise SystemExit(f"hipcc failed ({tag})")
    objs.append(obj)
r = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
if r.returncode != 0:
    raise SystemExit("link failed: " + r.stderr)
print(out)
