#!/usr/bin/env python3
"""Build a VARIANT of libgnm_hip.so next to the product library, for A/B timing on the GPU box
(select with GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/<name>.so):

    python tools/build_variant.py tuning -DGNM_AGG16_TUNING          # current sources + defines
    python tools/build_variant.py r01 --rev 1ce5467                  # the kernels of an older commit

Variants are built here (hipcc cross-compiles gfx950) and travel to the GPU box as git-ignored .so files."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
from gnm import _build  # noqa: E402


def main():
    name = sys.argv[1]
    defines = [a[2:] for a in sys.argv[2:] if a.startswith("-D")]
    rev = sys.argv[sys.argv.index("--rev") + 1] if "--rev" in sys.argv else None
    out = os.path.join(_build.LIB_DIR, "variants", name + ".so")
    with tempfile.TemporaryDirectory() as tmp:
        csrc = None
        if rev:
            csrc = os.path.join(tmp, "csrc")
            os.makedirs(csrc)
            files = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "--name-only", rev,
                                             "graph-neural-mapping_amd/csrc/"], text=True).split()
            for f in files:
                data = subprocess.check_output(["git", "-C", ROOT, "show", "%s:%s" % (rev, f)])
                open(os.path.join(csrc, os.path.basename(f)), "wb").write(data)
        print(_build.build(force=True, out=out, defines=defines, csrc=csrc, obj_dir=os.path.join(tmp, "obj")))


if __name__ == "__main__":
    main()
