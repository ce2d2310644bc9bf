"""Static checks on the gfx950 code objects (CPU container: hipcc cross-compiles, nothing runs on a GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scale-equivariant-imaging_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_untracked_loads_of_the_token_streaming_kernels_are_not_touched_before_their_wait(tmp_path):
    """token_gemm.hip loads the auxiliary rows of a tile with inline-asm global loads that the compiler does not track and
    waits for them by count. Between such a load and the counted s_waitcnt in front of its first use no
    instruction may read or write the destination registers -- a register copy the allocator placed there would move data
    that has not landed (this happened once: the copy sat in front of a wait that was tied to the registers). Also: no
    variant spills to scratch (a spill of such a register would be the same bug)."""
    asm = tmp_path / "token_gemm.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                    "-I", CSRC, os.path.join(CSRC, "token_gemm.hip"), "-o", str(asm)], check=True, capture_output=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_loads.py"), str(asm)], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stdout
    assert out.stdout.count("async destination registers") >= 9 and "tokgrad_regs_kernel" in out.stdout, out.stdout
    text = asm.read_text()
    import re
    scratch = {m.group(1): int(m.group(2)) for m in
               re.finditer(r"\.name:\s+(\S*(?:rowgemm|tokgrad|tokgrad_regs)_kernel\S*)\n\s+\.private_segment_fixed_size:\s+(\d+)", text)}
    assert scratch and all(v == 0 for v in scratch.values()), {k: v for k, v in scratch.items() if v}
