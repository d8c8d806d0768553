"""Builds libtimemachine_amd.so for gfx950 with hipcc (in-tree, next to the sources).

    python -m timemachine_amd.csrc.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to the GPU box
with the repo snapshot.  One object per translation unit, compiled in parallel, linked into one shared library.
Then the compiled Python module timemachine_amd/lib/custom_ops.<abi>.so (pybind11, wrap_custom_ops.cpp, g++) is
built against it.
-ffp-contract=off: the per-pair math must compile to the same instruction sequence at every call site (exclusions are
subtracted in fixed point and have to cancel bit-for-bit); every fused multiply-add we want is written explicitly.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtimemachine_amd.so")
# the compiled Python module (pybind11, host-only C++): timemachine_amd/lib/custom_ops.<abi>.so, linked against LIB
BINDING_SRC = "wrap_custom_ops.cpp"
BINDING = os.path.join(os.path.dirname(HERE), "lib", "custom_ops" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
CXX = os.environ.get("CXX", "g++")
SOURCES = ["nonbonded.hip", "bonded.hip", "fused.hip", "barostat.hip", "integrator.hip", "local_md.hip", "potential.hip", "c_api.cpp"]
HEADERS = [
    "common.hpp", "engine.hpp", "fixed_point.hip.hpp", "nb_pair.hip.hpp", "kernels_nonbonded.hip.hpp", "kernels_nonbonded_rowblock.hip.hpp", "kernels_nblist.hip.hpp", "kernels_bonded.hip.hpp", "philox.hip.hpp", "nb_math.hip.hpp", "nb_math_coeffs.h", "nb_es_table.hip.hpp", "nb_snapshot_test.hip.hpp",
    "profiler.hpp", "../../include/timemachine_amd.h",
]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
    "-Wno-inline-asm",  # the compaction asm of the tile kernel writes exec on purpose ("clobber list contains reserved registers")
]


def _stamp():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + [BINDING_SRC, "build.py"]:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(HERE, os.path.splitext(src)[0] + ".o")
    cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return src, obj, r.returncode, r.stdout + r.stderr


def build_variant(tag, defines, only=None):
    """debug/ablation variant: lib<...>_<tag>.so with extra -D flags (selected at run time with TM_AMD_LIB).
    `only`: recompile just these sources with the flags and link the product build's objects for the rest (tile-kernel
    experiments: the macros only reach nonbonded.hip)."""
    out = os.path.join(HERE, f"libtimemachine_amd_{tag}.so")
    objs, temps = [], []
    for src in SOURCES:
        if only is not None and src not in only:
            objs.append(os.path.join(HERE, os.path.splitext(src)[0] + ".o"))
            continue
        obj = os.path.join(HERE, os.path.splitext(src)[0] + f".{tag}.o")
        cmd = [HIPCC] + FLAGS + [f"-D{d}" for d in defines] + ["-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError(f"hipcc failed on {src}")
        objs.append(obj)
        temps.append(obj)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed")
    for o in temps:
        os.remove(o)
    return out


def build_binding(verbose=True):
    """custom_ops.<abi>.so: g++ only (no device code in it); RUNPATH $ORIGIN/../csrc finds the library it binds."""
    import pybind11

    cmd = [
        CXX, "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
        "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], os.path.join(HERE, BINDING_SRC), "-o", BINDING,
        "-L" + HERE, "-ltimemachine_amd", "-Wl,--enable-new-dtags,-rpath,$ORIGIN/../csrc",
    ]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose and (r.stdout + r.stderr).strip():
        print(f"--- {BINDING_SRC} ---\n{r.stdout + r.stderr}", file=sys.stderr)
    if r.returncode != 0:
        raise RuntimeError(f"{CXX} failed on {BINDING_SRC}")
    return BINDING


def build(force=False, verbose=True):
    stamp_file = os.path.join(HERE, ".build_stamp")
    stamp = _stamp()
    if (not force and os.path.exists(LIB) and os.path.exists(BINDING) and os.path.exists(os.path.join(HERE, "libtimemachine_amd_rowblock.so"))
            and os.path.exists(stamp_file) and open(stamp_file).read() == stamp):
        return LIB
    objs = []
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        for src, obj, rc, out in ex.map(_compile, SOURCES):
            if verbose and out.strip():
                print(f"--- {src} ---\n{out}", file=sys.stderr)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src}")
            objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stdout + r.stderr, file=sys.stderr)
        raise RuntimeError("link failed")
    build_binding(verbose)
    # the parity tests' variant library: the product objects + nonbonded.hip once more with the row-block kernel compiled in
    # (a second, independent implementation of the tile kernel that the GPU suite compares bit for bit; not in the product)
    build_variant("rowblock", ["TM_ROWBLOCK"], only=("nonbonded.hip",))
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv or "--nb-variant" in sys.argv:
        # --variant TAG DEFINES...: every source recompiled; --nb-variant TAG DEFINES...: nonbonded.hip only
        nb_only = "--nb-variant" in sys.argv
        i = sys.argv.index("--nb-variant" if nb_only else "--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2 :], only=("nonbonded.hip",) if nb_only else None))
    else:
        print(build(force="--force" in sys.argv))
