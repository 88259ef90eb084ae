"""The decoder used for the decoder-side check has to be a checker with teeth: damaged streams must not decode to the fixture's pictures, and the reference-encoder defect R1
(tests/decoder_check.py) must be what the `--ref-deblock-qp` switch says it is where the switch can know the QP.  Runs on the CPU: the streams come from the one-lane checker
build, which tests/test_stream_cpu.py holds to the reference's bytes."""
import random

import pytest

import decoder_check
from test_stream_cpu import GOLD, cpu, encode  # noqa: F401  (cpu: the fixture)


@pytest.fixture(scope="module")
def small(cpu):  # noqa: F811
    stream, recon, _ = encode(cpu, "200x136")
    assert recon == GOLD["200x136"]["recon_md5"]
    return stream


def test_decodes_to_the_reference_reconstruction(small):
    md5s, info = decoder_check.decode(small, 200, 136)
    assert md5s == GOLD["200x136"]["recon_md5"]
    assert info["pictures"] == 3 and info["substreams"] == 9 and info["entry_points"] == 6


def test_truncated_stream_is_refused(small):
    with pytest.raises(AssertionError, match="hevcdec exit code"):
        decoder_check.decode(small[:-7], 200, 136)


def test_damaged_slice_data_never_decodes_to_the_same_pictures(small):
    # flip one bit at 40 places inside the slice data of the pictures: the decoder either finds a violation (a sub-stream that does not end where its entry point says, a
    # terminating bin in the wrong place, a value out of range) or reconstructs different pictures - it never agrees with the fixture
    rng = random.Random(5)
    start = small.find(b"\x00\x00\x01\x26") + 12      # first IDR slice NAL unit, behind its header bytes
    assert start > 12
    caught = differs = 0
    for _ in range(40):
        pos = rng.randrange(start, len(small) - 2)
        bad = bytearray(small)
        bad[pos] ^= 1 << rng.randrange(8)
        try:
            md5s, _ = decoder_check.decode(bytes(bad), 200, 136)
        except AssertionError:
            caught += 1
            continue
        assert md5s != GOLD["200x136"]["recon_md5"]
        differs += 1
    assert caught >= 20, (caught, differs)      # (most single-bit errors derail the arithmetic decoder before the sub-stream ends)


@pytest.mark.parametrize("case", ["416x240_cbr300_eng2", "416x240_cbr400_perf1_eng2_wpp_rows"])
def test_rate_control_drift_is_the_deblocking_qp(cpu, case):  # noqa: F811
    g = GOLD[case]
    stream, recon, _ = encode(cpu, case)
    assert recon == g["recon_md5"]
    assert decoder_check.decode(stream, g["width"], g["height"])[0] != g["recon_md5"]
    assert decoder_check.decode(stream, g["width"], g["height"], ref_deblock_qp=True)[0] == g["recon_md5"]
