"""Toolchain regressions the product works around (kept as probes so that a compiler update that changes the picture is noticed)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_int16_register_array_rule(tmp_path):
    """profiles/r04_history.md: a lane's samples read into an int16_t register array before the first store gave wrong streams on the device; 32-bit temporaries are
    correct, and the product keeps 16-bit values out of register arrays (enc_prims.h intra_fill_refs).  The 32-bit form MUST be right; whether the 16-bit form of this
    small probe goes wrong with the installed compiler is reported, not required (the failure was seen inside the 560 KB encoder kernel)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "probe"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", os.path.join(HERE, "probes", "i16_regarray_probe.hip"), "-o", str(exe)], timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    fields = dict(kv.split("=") for kv in out.stdout.split())
    assert int(fields["int32_temporaries_wrong"]) == 0
    print("int16 register array form:", fields["int16_array_wrong"], "entries wrong")


def test_no_int16_register_arrays_in_the_encoder_core():
    """the rule itself, as a source check: no local `int16_t name[small constant]` arrays in the SPMD encoder core (struct members and pointers are fine)"""
    import re
    root = os.path.join(os.path.dirname(HERE), "homerhevc_amd", "csrc", "enc")
    bad = []
    for fn in sorted(os.listdir(root)):
        depth = 0
        for ln, line in enumerate(open(os.path.join(root, fn)), 1):
            code = line.split("//")[0]
            m = re.search(r"^\s+(?:const\s+)?(?:int16_t|uint16_t)\s+\w+\[(\d+)\]\s*(?:=|;)", code)
            # function bodies are indented with tabs; struct members sit one tab deep inside `struct X {`
            if m and int(m.group(1)) <= 16 and code.startswith("\t\t"):
                bad.append(f"{fn}:{ln}: {code.strip()}")
    assert not bad, "\n".join(bad)
