"""ISA-level guard (CPU): the gfx950 code objects inside libgnm_hip.so must not contain the instruction form that
miscompiled in round 2 -- a 16-byte (or 12-byte) `buffer_store` whose scalar offset is an SGPR.  With that form
hipcc (ROCm 7.2) schedules a VALU write of the store's data registers straight behind the store (its hazard table
exempts it) and gfx950 then stores the NEW value in some lanes (csrc/linear.hip, gnm_lin_stream_kernel: address
integers appeared in Z).  The sources avoid it by carrying the row offset in the VECTOR operand; this test makes a
compiler or source change that re-introduces the form fail on the CPU box, before any numeric test has to catch it."""
import glob
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graph-neural-mapping_amd", "lib",
                   "libgnm_hip.so")
WIDE_STORE = re.compile(r"\bbuffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*(?:v\d+|off),\s*(?:s\[\d+:\d+\]|ttmp\[\d+:\d+\]),\s*(\S+)")


def disassemble(tmp_path):
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    if not os.path.exists(LIB):
        from gnm import _build
        _build.build()
    so = shutil.copy(LIB, str(tmp_path))          # --offloading extracts the bundles NEXT to the file it is given
    subprocess.run([objdump, "--offloading", so], check=True, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    objs = sorted(glob.glob(so + ".*gfx950"))
    assert objs, "no gfx950 code object inside libgnm_hip.so"
    text = []
    for o in objs:
        text.append(subprocess.run([objdump, "-d", o], check=True, stdout=subprocess.PIPE, text=True).stdout)
    return "\n".join(text)


def test_no_wide_buffer_store_with_sgpr_soffset(tmp_path):
    asm = disassemble(tmp_path)
    stores = WIDE_STORE.findall(asm)
    assert len(stores) >= 50, "the disassembly no longer shows the kernels' 16-byte buffer stores (%d found)" % len(stores)
    bad = [s for s in stores if s.rstrip(",") != "0"]
    assert not bad, "16/12-byte buffer_store with a non-zero scalar offset (%s): the gfx950 store/VALU hazard of " \
                    "csrc/linear.hip gnm_lin_stream_kernel -- carry the offset in the vector operand" % sorted(set(bad))


def test_kernels_use_the_matrix_cores_and_no_scratch(tmp_path):
    """the same disassembly as a build check: bf16 and fp32 MFMA are present (aggregation / Linear kernels)."""
    asm = disassemble(tmp_path)
    assert "v_mfma_f32_32x32x16_bf16" in asm
    assert "v_mfma_f32_32x32x2_f32" in asm or "v_mfma_f32_32x32x2f32" in asm
