"""The traversal kernels must not use scratch memory: the persistent kernels count on their whole grid being resident at
once (one block per CU), and a kernel with scratch -- even the two bytes of a flag whose address is taken -- runs with
whatever the runtime's scratch pool allows at that moment (measured: the same launch 1.4 ms or 3.9 ms).  Compiles the kernel
source for gfx950 with the resource remarks on and reads them."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "volumerenderer_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_no_kernel_uses_scratch_or_spills_vector_registers(tmp_path):
    flags = re.search(r"^FPFLAGS\s*:=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1).split()
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only",
           "-o", str(tmp_path / "k.s"), os.path.join(CSRC, "vr_kernels.hip"), "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    names = re.findall(r"Function Name: (\S+)", out)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out)]
    vspill = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", out)]
    assert len(names) >= 15 and len(names) == len(scratch) == len(vspill)
    bad = [(n, s, v) for n, s, v in zip(names, scratch, vspill) if s or v]
    assert not bad, bad
