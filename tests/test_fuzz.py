"""Random configurations of the frame encoder against the compiled reference (tools/encoder_fuzz.py): picture sizes, clip seeds, QPs, rate-control targets, RD modes,
transform depths, thread and engine counts, scene cuts drawn from a fixed seed.  The checker build here, the device encoder under -m gpu (oracle/_ref/ travels to the
GPU box as built files; where it is missing the tests skip).  Longer runs of the same tool: profiles/r04_encoder_fuzz.md."""
import os
import subprocess
import sys

import pytest

import encoder_cases as ec
import libs

TOOL = os.path.join(ec.ROOT, "tools", "encoder_fuzz.py")
REF = os.path.join(libs.ORACLE_DIR, "_ref", "ref_ctudump")


def run(args, timeout, retry=True):
    r = subprocess.run([sys.executable, TOOL] + args, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln and not ln.startswith(" ")]
    # (a differing case that had evaluations on a stale prediction window is the documented exception, include/homer_gpu.h: hmr_gpu_enc_stale_predictions)
    bad = [ln for ln in lines if "IDENTICAL" not in ln and "REFUSED" not in ln and "quirk Q12" not in ln and "differing cases" not in ln]
    # a differing case gets a second run on its own: the turnstiled reference has been seen to differ from itself once in about 1900 runs with several engines
    # (profiles/r04_encoder_fuzz.md, finding 7); a real difference repeats
    if bad and retry:
        flags = ["--gpu"] if "--gpu" in args else []
        if "--chain-sets" in args:
            flags += ["--chain-sets", args[args.index("--chain-sets") + 1]]
        again = run(flags + [ln.split()[0] for ln in bad], timeout, retry=False)
        bad = [ln for ln in again if "IDENTICAL" not in ln]
    assert not bad, "\n".join(bad[:5]) + r.stderr[-500:]
    return lines


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (the reference's sources are only in the build container)")
def test_checker_build_on_random_configurations():
    lines = run(["--cases", "16", "--seed", "101", "--max-ctus", "40"], 1200)
    assert sum("IDENTICAL" in ln for ln in lines) >= 15


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
def test_device_encoder_on_random_configurations():
    lines = run(["--gpu", "--cases", "90", "--seed", "102", "--max-ctus", "120"], 900)
    assert sum("IDENTICAL" in ln for ln in lines) >= 80


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
def test_device_batch_call_on_random_configurations():
    """eight random cases per hmr_gpu_enc_encode_batch call: CTUs of pictures of different sizes and settings share the pool's workers"""
    lines = run(["--gpu", "--batch", "8", "--cases", "90", "--seed", "103", "--max-ctus", "120"], 900)
    assert sum("IDENTICAL" in ln for ln in lines) >= 45


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
def test_device_chain_call_on_random_configurations():
    """several engines through hmr_gpu_enc_encode_chain, two objects per engine: overlapped frames of one sequence"""
    lines = run(["--gpu", "--engines-only", "--chain-sets", "2", "--cases", "30", "--seed", "104", "--max-ctus", "150", "--max-cols", "16"], 900)
    assert sum("IDENTICAL" in ln for ln in lines) >= 28
