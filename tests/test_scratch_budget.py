"""The encoder kernel's private memory is a budget, not an accident (profiles/r05_history.md): register spills, call frames and local arrays indexed at run
time were half of the kernel's fabric writes and a load in front of every out-of-line primitive until round 5 took them out.  The cross-compile for gfx950 reports what the
kernel reserves per lane and how many vector registers it spills; this test holds the line (no GPU needed: hipcc cross-compiles here, about a minute)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_encoder_kernel_private_memory_budget(tmp_path):
    out = tmp_path / "k.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-S", "-o", str(out), os.path.join(ROOT, "homerhevc_amd", "csrc", "k_encode.hip")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = {}
    for blk in r.stderr.split("Function Name: ")[1:]:
        name = blk.split()[0]
        f = {k: int(v) for k, v in re.findall(r"(ScratchSize \[bytes/lane\]|VGPRs Spill|VGPRs): (\d+)", blk)}
        seen[name] = f
    pools = [v for k, v in seen.items() if "k_encode_pool" in k]      # the throughput kernel and the latency kernel (k_encode_pool_lat)
    assert len(pools) == 2
    pool = max(pools, key=lambda v: v["ScratchSize [bytes/lane]"])
    # round 4: 2080 bytes and 53 spilled registers; round 5's last build: 760 and 1
    assert pool["ScratchSize [bytes/lane]"] <= 1024, pool
    assert pool["VGPRs Spill"] <= 8, pool
    assert pool["VGPRs"] <= 256, pool
    asm = out.read_text()
    # the matrix-core transforms are in the kernel (v_mfma_f32_16x16x16_f16), and the motion search's candidate arrays are not indexed through private memory
    assert asm.count("v_mfma_f32_16x16x16_f16") >= 50
    body = asm[asm.index("_ZN4henc16motion_inter_ctu"):]
    body = body[:body.index(".Lfunc_end")]
    variable_scratch = [l for l in body.splitlines() if "scratch_" in l and "Spill" not in l and "Reload" not in l]
    assert len(variable_scratch) <= 16, variable_scratch[:10]      # (the merge evaluation's two five-byte lists; the search's candidate arrays alone were 35)
