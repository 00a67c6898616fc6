"""Build libgnm_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
# GNM_HIP_LIB: load another build of the same C-ABI instead (A/B timing of kernel variants made by
# tools/build_variant.py); unset in normal use
LIB_PATH = os.environ.get("GNM_HIP_LIB") or os.path.join(LIB_DIR, "libgnm_hip.so")
SOURCES = ["agg.hip", "aggm.hip", "maxpool.hip", "linear.hip", "norm.hip", "disc.hip", "head.hip", "tail.hip", "sgemm.hip", "evalfwd.hip", "evallayer.hip", "host.cpp"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libgnm_hip.so cannot be built")


def needs_build():
    if os.environ.get("GNM_HIP_LIB"):
        return False                    # an explicitly chosen variant is used as it is
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None, defines=(), csrc=None, obj_dir=None):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC csrc/* -> lib/libgnm_hip.so
    (out / defines / csrc / obj_dir: variant builds of tools/build_variant.py)"""
    if out is None and not force and not needs_build():
        return LIB_PATH
    target = out or os.path.join(LIB_DIR, "libgnm_hip.so")
    src_dir = csrc or CSRC
    os.makedirs(os.path.dirname(target), exist_ok=True)
    objs = []
    obj_dir = obj_dir or os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(src_dir, src), "-o", obj,
               "-Wno-pass-failed"] + ["-D" + d for d in defines]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
    tmp = target + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout))
    os.replace(tmp, target)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
