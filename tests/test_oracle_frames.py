"""Frame-level oracle (deblock, SAO statistics, SAO offset, padding) against golden vectors minted by running the
reference's own in-loop code with the reference encoder's own side-info (tests/golden/make_golden_frames.py)."""
import numpy as np

import frame_cases as fc
import golden_io


def _check(got, exp, W, H):
    u4, w4 = H // 4, W // 4
    a, b = got["bs_ver"][:u4, :w4].copy(), exp["bs_ver"][:u4, :w4].copy()
    a[:, 1::2] = 0; b[:, 1::2] = 0
    assert np.array_equal(a, b)
    a, b = got["bs_hor"][:u4, :w4].copy(), exp["bs_hor"][:u4, :w4].copy()
    a[1::2, :] = 0; b[1::2, :] = 0
    assert np.array_equal(a, b)
    for k in ("deblocked", "sao", "padded"):
        for i in range(3):
            assert np.array_equal(got[k][i], exp[k][i]), (k, i)
    assert np.array_equal(got["stats"], exp["stats"])


def test_oracle_frames_match_reference_goldens(oracle):
    cases = golden_io.load_frame_goldens()
    assert len(cases) == 4
    for case, exp, meta in cases:
        got = fc.run_oracle(oracle, case)
        _check(got, exp, case["width"], case["height"])
        # the fixture must exercise the filters, not just pass through
        assert (got["deblocked"][0] != case["pre"][0]).sum() > 2000
        assert (got["sao"][0] != got["deblocked"][0]).sum() > 500


def test_zscan_to_raster_matches_reference_table(oracle, ref):
    """a3: per-CTU z-order <-> raster (abs2raster_table, hmr_encoder_lib.c:95-100)."""
    import ctypes as C
    import numpy as np
    import pytest
    if ref is None:
        pytest.skip("oracle/_ref not built")
    t = np.zeros(256, np.int32)
    ref.refh_abs2raster(t.ctypes.data_as(C.c_void_p))
    assert [oracle.ora_zscan_to_raster(a) for a in range(256)] == t.tolist()
