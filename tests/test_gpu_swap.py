"""Swap test (SURVEY.md §8-V acceptance harness): the reference's own decision code + CABAC, compiled in the build
container into oracle/_ref, encodes a clip with every low_level_funcs_t entry (and the directly-called sad /
fill_reference_samples) routed to the HIP kernels through the drop-in C ABI.  The .265 must be byte-identical to
the unswapped run of the same binary build."""
import os
import subprocess
import sys

import pytest

import libs

pytestmark = pytest.mark.gpu

SWAP = os.path.join(libs.ORACLE_DIR, "_ref", "ref_swap")


def _encode(exe, clip, out, w, h, frames, env=None, extra=()):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, clip, out, str(w), str(h), str(frames), *extra], capture_output=True, text=True, env=e, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(out, "rb") as f:
        return f.read(), r.stderr


@pytest.mark.parametrize("extra,w,h,frames", [((), 200, 136, 3), (("force_intra=1",), 200, 136, 2), ((), 416, 240, 3), (("rd=1",), 200, 136, 2),
                          (("bitrate_mode=1", "bitrate=400", "perf=1"), 416, 240, 3),                       # BASELINE configs[2] shape: CBR, performance_mode 1
                          (("force_intra=1", "rd=1", "intra_tr=4", "perf=0"), 200, 136, 2)],               # BASELINE configs[4] shape: all-intra, full RDO, TU depth 4
                         ids=["ippp_200x136", "all_intra_200x136", "ippp_416x240", "ippp_rdfull_200x136", "cbr_perf1_416x240", "all_intra_rdfull_tr4_200x136"])
def test_stream_identical_with_gpu_kernels(tmp_path, extra, w, h, frames):
    if not (os.path.exists(SWAP) and os.path.exists(libs.REF_LOCKSTEP)):
        pytest.skip("oracle/_ref not shipped (built only where the reference sources exist)")
    sys.path.insert(0, os.path.join(libs.ROOT, "tools"))
    import gen_yuv
    clip = str(tmp_path / "clip.yuv")
    gen_yuv.write_clip(clip, w, h, frames)
    ref, _ = _encode(libs.REF_LOCKSTEP, clip, str(tmp_path / "ref.265"), w, h, frames, extra=extra)
    same, _ = _encode(SWAP, clip, str(tmp_path / "none.265"), w, h, frames, env={"HOMER_SWAP": "none"}, extra=extra)
    assert same == ref, "harness itself changes the stream"
    gpu, log = _encode(SWAP, clip, str(tmp_path / "gpu.265"), w, h, frames, env={"HOMER_SWAP": "all"}, extra=extra)
    rd_full = "rd=1" in extra
    # with full RDO the CABAC bit estimate enters the tree comparison, so the luma CU driver stays on the host and its two halves are routed instead
    drivers = ("intra mode search routed", "intra TU chain routed") if rd_full else ("luma intra CU driver routed", "chroma intra CU driver routed")
    if not any(e.startswith("force_intra") for e in extra):
        drivers += ("inter CU transform tree routed",)
    for what in ("table entries routed", "deblocking routed", "border padding routed", "SAO offset derivation routed") + drivers:
        assert what in log, (what, log[-600:])
    assert len(ref) > 300
    assert gpu == ref, f"stream differs: {len(gpu)} vs {len(ref)} bytes"
    if not rd_full:    # the same encode with the CU driver left to the host: search and TU chain are then routed call by call
        low, log = _encode(SWAP, clip, str(tmp_path / "low.265"), w, h, frames, env={"HOMER_SWAP": "all,-intra_luma_cu,-intra_chroma_cu,-inter_cu"}, extra=extra)
        assert "luma intra CU driver routed" not in log and "chroma intra CU driver routed" not in log and "inter CU transform tree routed" not in log
        for what in ("intra mode search routed", "intra TU chain routed"):
            assert what in log, (what, log[-600:])
        assert low == ref


def test_wpp_threads_share_the_gpu(tmp_path):
    """The reference calls its table from several WPP threads at once; the drop-in entries must serialise on the shared context.  Forced-intra
    streams do not depend on the thread count, so 4 threads driving the GPU kernels must reproduce the 1-thread reference."""
    if not (os.path.exists(SWAP) and os.path.exists(libs.REF_LOCKSTEP)):
        pytest.skip("oracle/_ref not shipped (built only where the reference sources exist)")
    sys.path.insert(0, os.path.join(libs.ROOT, "tools"))
    import gen_yuv
    w, h, frames = 416, 240, 2
    clip = str(tmp_path / "clip.yuv")
    gen_yuv.write_clip(clip, w, h, frames)
    ref1, _ = _encode(libs.REF_LOCKSTEP, clip, str(tmp_path / "r1.265"), w, h, frames, extra=("force_intra=1",))
    ref4, _ = _encode(libs.REF_LOCKSTEP, clip, str(tmp_path / "r4.265"), w, h, frames, extra=("force_intra=1", "wpp=4"))
    assert ref4 == ref1, "reference itself depends on the thread count"
    gpu4, _ = _encode(SWAP, clip, str(tmp_path / "g4.265"), w, h, frames, env={"HOMER_SWAP": "all"}, extra=("force_intra=1", "wpp=4"))
    assert gpu4 == ref1
