"""Build-time guard (no GPU): no kernel of the HIP library may spill to scratch memory.

A spilled address or operand turns one instruction into a round trip through memory with a full `vmcnt` wait — twice in
round 2 that was most of a kernel's time (profiles/r02_gproj.md) — and nothing else reports it: the kernels stay correct.
The check reads the AMDGPU metadata note of every object the Makefile built (`.private_segment_fixed_size`,
`.vgpr_spill_count`) with the LLVM tools of the ROCm image."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]


def _kernels(obj, tmp):
    base = os.path.join(tmp, os.path.basename(obj))
    subprocess.run([TOOLS[0], f"--dump-section=.hip_fatbin={base}.fat", obj], check=True, capture_output=True)
    subprocess.run([TOOLS[1], "--unbundle", "--type=o", f"--input={base}.fat", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={base}.co"], check=True, capture_output=True)
    notes = subprocess.run([TOOLS[2], "--notes", f"{base}.co"], check=True, capture_output=True, text=True).stdout
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", k).group(1))
        spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", k)
        yield name, scratch, int(spill.group(1)) if spill else 0


def test_no_kernel_spills_to_scratch(tmp_path):
    objs = sorted(glob.glob(os.path.join(ROOT, "analysisgnn_amd", "csrc", "build", "*.hip.o")))
    if not objs or not all(os.path.exists(t) for t in TOOLS):
        pytest.skip("objects not built here or LLVM tools missing")
    seen, bad = 0, []
    for obj in objs:
        for name, scratch, spill in _kernels(obj, str(tmp_path)):
            seen += 1
            if scratch or spill:
                bad.append((os.path.basename(obj), name, scratch, spill))
    assert seen > 100, f"only {seen} kernels found: the metadata parser no longer matches the tool's output"
    assert not bad, f"kernels using scratch memory (object, kernel, bytes, spilled VGPRs): {bad}"
