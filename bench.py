#!/usr/bin/env python3
"""bench.py - encoded frames/sec of the device encoder on BASELINE.json configs[1].

Workload ("cfg2-1080p-encode"): the real encode of 1920x1080 IPPP sequences (gop_size=1, fixed QP 32, quarter-pel ME, SAO on, one WPP thread per CTU row) through the
C ABI (include/homer_gpu.h section 12) - a batch of --sequences independent sequences per GPU (256; eight different synthetic clips, tools/gen_yuv.py), one frame of
each per step (hmr_gpu_enc_encode_batch_pipelined).  One step = everything HOMER_enc_encode does for those frames: the phase planes of the reference pictures, then ONE
launch of k_encode_pool - the CTU decisions of all the pictures as a pool of tasks on four row workers per CU and, behind each CTU, its post-decision tasks: deblocking,
SAO statistics / decision / syntax / offsets, CABAC of the CTU row's sub-stream, border padding - then the download of the sub-streams and headers / entry points / NAL
escaping on the host; the access units are delivered inside the timed region (the pipelined call delivers a step's units with the next call: the pipeline is empty when
the region starts and flushed before it ends).  The source pictures are resident in HBM before the timed region starts.  Warm-up frames are the first frames of the
sequences (the I frame and the first P frames), the timed frames the P frames that follow - every frame depends on the reconstruction of the one before, nothing is
replayed or cached.  Every access unit of every sequence is checked against the compiled reference's per-frame digests of its clip (tests/golden/bench_md5.json:
`stream_matches_reference`, `frames_checked_against_reference`, `clips`) for any --steps / --warmup the fixtures cover (24-40 frames).  Beside the headline:
`single_sequence` (one sequence alone, frame by frame and with its engines overlapped in one launch: hmr_gpu_enc_encode_chain), `single_thread_order`
(wfpp_num_threads = 1, the reference's deterministic single-thread stream 2f0c3447...), `at_2160p`, `cfg3_2160p_cbr` (BASELINE configs[2]: rate control in the CTU
kernel), `cfg5_2160p_intra_rdfull` (configs[4]: all-intra, full RDO, intra TU depth 4).

Multi-GPU (--gpus N under torch.distributed.run): one engine per GPU, as BASELINE.json's north_star and configs[3] say - the reference's
num_enc_engines = N frame pipeline (encoder_engine_thread, hmr_encoder_lib.c:3043) with engine k on rank k.  Every rank keeps one engine object of every
sequence; frame t of sequence s is encoded on rank (s + t) mod N, and after each step the ranks pass the final pictures (8-bit, without margins) and
the frame-to-frame scalars round the ring in ONE packed RCCL send / recv per rank (homerhevc_amd/engines.py).  N x min(--sequences, 160) sequences are in flight, so
every rank encodes that many frames per step whatever N is (weak scaling); the exchange is inside the timed region.  Every access unit is checked against
the reference's num_enc_engines = N stream (oracle/ref_ctudump.c's engine turnstile; tests/golden/bench_md5.json).

Extra objects: `roofline` for k_encode_pool (98 % of a step): algorithmic bytes per SURVEY.md 8-d against the 8 TB/s HBM peak, the fabric traffic from the
calibrated counter passes of this build, and `issue_bound` - what actually binds the kernel (the issue latency of two wavefronts per SIMD and the memory trips of
their dependent chains), with the ceiling measured live by an issue-rate probe; `roofline.subpel_planes`: the bandwidth-bound phase-plane kernels measured live;
`cpu_baseline`: the compiled reference (oracle/_ref/ref_lockstep) timed on this host on the same configuration by its own clock - one process per physical core
side by side (the headline of the baseline), one process, and one process with a thread per CTU row.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

# one hardware queue per concurrent encoder instance of the multi_stream measurement (the default of 4 serialises kernels of streams that share a queue);
# must be set before the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s
WORKERS_PER_CU = 4             # k_encode_pool: row workers (a wavefront + its helper, 40.6 KB of LDS) a CU holds (k_encode.hip workers_per_cu)
WORKLOADS = {                  # name -> (width, height, configuration keys of tests/encoder_cases.default_cfg)
    # BASELINE.json configs[1] "WPP CTU rows on-GPU": one WPP thread per CTU row (wfpp_num_threads = 17), the reference's synchronous-wavefront schedule
    "cfg2-1080p-encode": (1920, 1080, {"wpp": 17}),
    # the same encode in the reference's single-thread order (wfpp_num_threads = 1): guesses + verification passes (enc_sched.h)
    "cfg2-1080p-encode-single-thread-order": (1920, 1080, {}),
    "cfg2-2160p-encode": (3840, 2160, {"wpp": 32}),                          # the same encode at 2160p (configs[3] per engine): 34 CTU rows on the reference's maximum of 32 threads
    "cfg2-416x240-encode": (416, 240, {"wpp": 4}),                           # quick look
    # BASELINE.json configs[2]: 2160p IPPP, CBR 20000 kbps (vbv = 1 s, 35 % initial fullness), performance_mode 1 - the rate control runs in the CTU kernel
    "cfg3-2160p-cbr": (3840, 2160, {"wpp": 32, "bitrate_mode": 1, "bitrate": 20000, "perf": 1}),
    "cfg3-1080p-cbr": (1920, 1080, {"wpp": 17, "bitrate_mode": 1, "bitrate": 5000, "perf": 1}),
    # BASELINE.json configs[4] as BASELINE.md realises it: 2160p all-intra (every picture an I picture), rd_mode 1 = full RDO (the CABAC bit counter prices every intra
    # decision), performance_mode 0, intra TU depth 4
    "cfg5-2160p-intra-rdfull": (3840, 2160, {"wpp": 32, "force_intra": 1, "rd": 1, "intra_tr": 4, "perf": 0}),
    "cfg5-1080p-intra-rdfull": (1920, 1080, {"wpp": 17, "force_intra": 1, "rd": 1, "intra_tr": 4, "perf": 0}),
}
# the sequences of a batch encode eight different clips (tools/gen_yuv.py: 1234 is the published clip); the reference's digests of each are in bench_md5.json
CLIP_SEEDS = [1234, 1, 2, 3, 4, 5, 6, 7]


def seed_workload(workload, seed):
    return workload if seed == 1234 else f"{workload}-seed{seed}"
# md5 of the reference's stream after every access unit, per workload (tests/golden/bench_md5.json, minted by tests/golden/make_bench_golden.py from the compiled
# reference: ref_lockstep for one thread, ref_ctudump under HOMER_TURNSTILE for one thread per row): whatever --steps / --warmup, the digest after frame k is checked
REFERENCE_MD5 = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_md5.json")))


def check_against_reference(workload, cumulative):
    """cumulative[k] = md5 of the produced stream after access unit k.  Returns (matches, frames checked): every frame the fixture covers must agree."""
    gold = REFERENCE_MD5.get(workload, {}).get("cumulative_md5", [])
    n = min(len(gold), len(cumulative))
    if n == 0:
        return False, 0
    return all(cumulative[k] == gold[k] for k in range(n)), n


def latest_profile(suffix):
    """The newest committed counter-pass summary profiles/rNN_<suffix> (NN = round; the `final` set of a round before its first)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_final_{suffix}")) + glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")),
                   key=lambda q: (os.path.basename(q)[:3], "_final_" in q))
    return found[-1] if found else None


def pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def headline_line(out):
    """The driver's record: ONE compact JSON line (well under 4 KB) - the contract's keys, `roofline` and `cpu_baseline` as objects of numbers and short strings, and one
    number per secondary section.  Everything else of `out` goes to earlier lines (emit)."""
    cfg = out.get("config", {})
    roof = out.get("roofline") or {}
    line = pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line["config"] = pick(cfg, "workload", "sequences_per_gpu", "sequences", "num_enc_engines", "width", "height", "wfpp_num_threads", "qp", "gop", "stream_matches_reference")
    line["config"]["frames_checked"] = cfg.get("frames_checked_against_reference", cfg.get("access_units_checked_against_reference"))
    r = pick(roof, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_build", "traffic_is_of_this_build", "ms_per_launch", "launches",
             "algorithmic_bytes_per_launch", "algorithmic_bytes_model")
    ib = roof.get("issue_bound") or {}
    if ib:
        r["issue_bound"] = pick(ib, "wave_instructions_per_frame", "salu_over_valu", "valu_issue_frac", "wait_share_of_wave_cycles", "valu_lane_utilisation", "build")
    sp = roof.get("subpel_planes") or {}
    if sp:
        r["subpel_planes"] = pick(sp, "achieved", "frac", "ms_per_picture")
    line["roofline"] = r or None
    cb = out.get("cpu_baseline")
    if cb:
        c = pick(cb, "value", "unit", "cores", "kind")
        c["sample"] = cb.get("sample", "")[:200]
        c["one_process"] = (cb.get("one_process") or {}).get("value")
        rows = cb.get("one_thread_per_ctu_row") or {}
        c["rows_threads"] = {"value": rows.get("value"), "threads": rows.get("threads")}
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    ss = out.get("single_sequence")
    if ss:
        line["single_sequence"] = {"value": ss.get("value"), "stream_matches_reference": ss.get("stream_matches_reference"),
                                   "engines_overlapped": [pick(lane, "workload", "frames_per_s_full_chains", "stream_matches_reference") for lane in ss.get("engines_overlapped", [])]}
    for name in ("single_thread_order", "serial_order_batch", "at_2160p", "cfg3_2160p_cbr", "cfg5_2160p_intra_rdfull"):
        sec = out.get(name)
        if isinstance(sec, dict):
            d = pick(sec, "value", "stream_matches_reference")
            d["sequences"] = sec.get("sequences") or (sec.get("config") or {}).get("sequences_per_gpu")
            d["alone"] = (sec.get("single_sequence") or {}).get("value")
            d["cpu"] = (sec.get("cpu_baseline") or {}).get("value")
            d["cpu_one_process"] = ((sec.get("cpu_baseline") or {}).get("one_process") or {}).get("value")
            line[name] = {k: v for k, v in d.items() if v is not None}
    line["stream_matches_reference"] = out.get("stream_matches_reference")
    line.update(pick(out, "access_units_checked_against_reference", "access_units_differing", "access_units_produced"))
    line["stale_prediction_windows"] = out.get("evaluations_on_a_stale_prediction_window")
    line["build"] = roof.get("this_build")
    return line


def emit(out):
    """stdout: the full record section by section on EARLIER lines (`{"detail": name, ...}`, also kept as gpurun_out/bench_detail.json where that directory can be written),
    then the compact headline as the LAST line."""
    head = headline_line(out)
    for name, sec in out.items():
        if isinstance(sec, (dict, list)):
            print(json.dumps({"detail": name, "content": sec}))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w"), indent=1)
    except OSError:
        pass
    text = json.dumps(head)
    assert len(text) < 4096, f"headline line grew to {len(text)} bytes"
    sys.stdout.flush()
    print(text, flush=True)


def load_lib():
    lib = C.CDLL(os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))   # no fallback: without the HIP library there is no bench
    import encoder_cases as ec
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_create_serial_pool.argtypes = lib.hmr_gpu_enc_create.argtypes
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long), C.c_char_p]
    lib.hmr_gpu_enc_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    return lib


def describe_keys(keys):
    """The configuration a WORKLOADS key set stands for, in the words of BASELINE.json (the `sample` text of a cpu_baseline)."""
    gop = "all-intra" if keys.get("force_intra") else "IPPP"
    rate = f"CBR {keys['bitrate']} kbps" if keys.get("bitrate_mode") == 1 else f"VBR {keys['bitrate']} kbps" if keys.get("bitrate_mode") else "QP32"
    extra = "".join(f" {k}={keys[k]}" for k in ("rd", "intra_tr", "perf") if k in keys)
    return f"{gop} {rate}{extra}"


def cpu_baseline(width, height, keys, frames):
    """The compiled reference, one thread (wpp = 1, engines = 1: the configuration the device output is identical to), on this host: one process, one process with a
    thread per CTU row, and K = host cores processes side by side (one sequence each, `frames` frames).  The times are the harness's own (oracle/ref_lockstep.c prints
    the seconds between the first HOMER_enc_encode and the last access unit): process start, HOMER_enc_init and the library's table set-up are NOT in them."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_lockstep")
    if not os.path.exists(exe):
        return None
    import gen_yuv

    def seconds(stdout):
        for line in reversed(stdout.decode("latin1").splitlines()):
            if line.startswith("LOCKSTEP"):
                f = dict(kv.split("=") for kv in line.split()[1:])
                return float(f["seconds"]), int(f["frames"])
        raise RuntimeError("no LOCKSTEP line from the reference harness")

    ncores = os.cpu_count() or 1
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 64 << 30
    # what this process may actually use of the host: its affinity mask and its cgroup's CPU quota
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(ncores))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            quota = open(path).read().strip()
            break
        except OSError:
            pass
    quota_cpus = None
    if quota and quota.split()[0] not in ("max", "-1"):
        f = quota.split()
        quota_cpus = round(int(f[0]) / (int(f[1]) if len(f) > 1 else 100000), 2)
    # one process per PHYSICAL core, pinned to it (the siblings of a core share its SSE units and caches: 256 processes on 128 cores were slower in total than 128)
    first_cpu_of_core = {}
    for cpu in allowed:
        try:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            key = (open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip())
        except OSError:
            key = ("?", str(cpu))
        first_cpu_of_core.setdefault(key, cpu)
    pins = sorted(first_cpu_of_core.values())
    K = max(1, min(len(pins), int(avail * 0.5 / (400 << 20))))      # (a 1080p reference process holds well under 400 MB)
    if quota_cpus:
        K = max(1, min(K, int(quota_cpus)))
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "in.yuv")
        gen_yuv.write_clip(yuv, width, height, frames)
        cmd = [exe, yuv, "-", str(width), str(height), str(frames)] + [f"{k}={v}" for k, v in keys.items()]

        def pinned(cpu):
            return lambda: os.sched_setaffinity(0, {cpu})

        dt, n1 = seconds(subprocess.run(cmd, check=True, capture_output=True, preexec_fn=pinned(pins[0])).stdout)
        rows = (height + 63) // 64
        try:      # (free-running threads: at 3840x2160 the reference has been seen to die with SIGSEGV in this mode - recorded, not fatal)
            dt_rows, _ = seconds(subprocess.run(cmd + [f"wpp={min(rows, 32)}"], check=True, capture_output=True, timeout=600).stdout)      # (MAX_NUM_THREADS 32, hmr_private.h:1234)
            rows_failed = None
        except (subprocess.CalledProcessError, subprocess.TimeoutExpired, RuntimeError) as ex:
            dt_rows, rows_failed = None, f"{type(ex).__name__}: returncode {getattr(ex, 'returncode', None)}"
        # the same shape as the batch: K independent sequences at once, one single-thread reference process per physical host core
        t0 = time.time()
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, preexec_fn=pinned(pins[k])) for k in range(K)]
        secs, cpu_s = [], []
        for p in procs:
            out = p.stdout.read()
            _, status, ru = os.wait4(p.pid, 0)
            p.returncode = os.waitstatus_to_exitcode(status)
            secs.append(seconds(out)[0])
            cpu_s.append(ru.ru_utime + ru.ru_stime)
        wall_k = time.time() - t0
    one = n1 / dt
    aggregate = K * frames / max(secs)
    effective = aggregate / one
    throughput = {"value": round(aggregate, 3), "unit": "frames/s aggregate", "processes": K, "cores": round(effective, 1), "host_cores": ncores, "host_physical_cores_allowed": len(pins),
                  "affinity_cpus": len(allowed), "cgroup_cpu_max": quota, "frames_per_process": frames,
                  "slowest_process_s": round(max(secs), 2), "fastest_process_s": round(min(secs), 2), "wall_s_incl_process_start": round(wall_k, 2),
                  "cpu_seconds_per_process_mean": round(sum(cpu_s) / len(cpu_s), 2), "cpu_seconds_per_process_max": round(max(cpu_s), 2),
                  "one_process_encode_s": round(dt, 2),
                  "parallel_efficiency": round(effective / K, 3),
                  "note": "K single-thread reference processes side by side, each pinned to its own physical core, one sequence each: the host's answer to a batch of independent sequences; "
                          "K x frames / the slowest process's encode time (start-up and HOMER_enc_init excluded).  `cores` is the EFFECTIVE parallelism (aggregate rate / the rate of one "
                          "process alone): cpu_seconds_per_process against one_process_encode_s says whether the processes got their cores (equal: they did, and the loss is the memory "
                          "system's; larger wall than CPU time: they were descheduled - a quota or other tenants)"}
    # the headline of the baseline is the host's best answer to the bench's workload (a batch of independent sequences): all cores, one reference process each
    return {"value": throughput["value"], "unit": "frames/s", "cores": throughput["cores"], "kind": "reference",
            "sample": f"{K} processes x {frames} frames {width}x{height} {describe_keys(keys)} through oracle/_ref/ref_lockstep (SSE4.2 table, wpp=1, engines=1), one pinned to each physical host core this "
                      f"process may use; the harness's own clock from the first HOMER_enc_encode to the last access unit of the slowest process (no process start, no HOMER_enc_init); `cores` = "
                      f"effective parallelism {round(effective, 1)} of {K} processes",
            "throughput": throughput,
            "one_process": {"value": round(one, 3), "unit": "frames/s", "cores": 1, "frames": n1},
            "one_thread_per_ctu_row": {"value": round(n1 / dt_rows, 3) if dt_rows else None, "threads": min(rows, 32), "host_cores": ncores, "failed": rows_failed,
                                       "note": "the same reference run free with wfpp_num_threads = CTU rows (its multi-thread mode; output depends on timing; `failed` when the reference "
                                               "process itself died or hung in this mode)"}}


def valu_issue_probe(lib, device):
    """hmr_gpu_probe_valu_issue (include/homer_gpu.h section 14): plain vector instructions per second of this device - all SIMDs busy with independent instructions
    (the ceiling the guide's 1024 SIMDs x clock / 4 cycles describes) and with ONE dependent chain per wavefront (what a row worker's decision chain is made of)."""
    lib.hmr_gpu_probe_issue.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    ctx = C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), device, None) == 0
    rows = []
    names = ["v_mad_u32_u24", "v_add_u32", "v_mov_b32", "v_perm_b32", "s_add_u32"]
    for op, waves, dep in ((1, 1, 0), (1, 2, 0), (1, 4, 0), (2, 2, 0), (0, 1, 0), (0, 2, 0), (0, 4, 0), (3, 2, 0), (1, 1, 1), (1, 2, 1), (0, 1, 1)):
        rate, ms = C.c_double(), C.c_double()
        assert lib.hmr_gpu_probe_issue(ctx, op, waves, dep, C.byref(rate), C.byref(ms)) == 0, lib.hmr_gpu_last_error()
        rows.append({"instruction": names[op], "waves_per_simd": waves, "dependent_chain": bool(dep), "wave_instructions_per_s": round(rate.value / 1e9, 2), "unit": "G/s", "ms": round(ms.value, 2),
                     "cycles_per_instruction_and_simd_at_2p4GHz": round(2.4e9 * 1024 / rate.value, 3)})
    lib.hmr_gpu_destroy(ctx)
    return {"kernel": "k_probe_valu (64 instructions of one kind per round, no memory access in the loop)", "rows": rows,
            "note": "independent instructions reach the guide's ceiling from one wavefront per SIMD on; a dependent chain issues one instruction per the ALU's latency - "
                    "k_encode_pool's workers are such chains, two wavefronts per SIMD (a worker and a helper, or two workers' wavefronts)"}


def subpel_planes_roofline(lib, torch, width, height, reps=20):
    """The frame-level kernels that replace every interpolation of the CTU walk (k_subpel.hip): one pass over the padded reference picture per P frame, 2 bytes in and
    16 (luma) / 64 (chroma) bytes out per sample - streaming work, timed here with events on the context's stream against the HBM peak."""
    lib.hmr_gpu_subpel_planes.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 3
    lib.hmr_gpu_stream.restype = C.c_void_p
    lib.hmr_gpu_stream.argtypes = [C.c_void_p]
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    ctx = C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), torch.cuda.current_device(), None) == 0
    sy, ry = ((width * 2 + 15) // 16 * 16) // 2 + 160, height + 160
    sc, rc = ((width // 2 * 2 + 15) // 16 * 16) // 2 + 80, height // 2 + 80
    pic = [torch.randint(0, 256, (r * st,), dtype=torch.int16, device="cuda") for st, r in ((sy, ry), (sc, rc), (sc, rc))]
    out = [torch.empty(16 * sy * ry, dtype=torch.uint8, device="cuda"), torch.empty(64 * sc * rc, dtype=torch.uint8, device="cuda"), torch.empty(64 * sc * rc, dtype=torch.uint8, device="cuda")]
    stream = torch.cuda.ExternalStream(lib.hmr_gpu_stream(ctx))
    args = [ctx] + [C.c_void_p(t.data_ptr()) for t in pic] + [sy, ry, sc, rc] + [C.c_void_p(t.data_ptr()) for t in out]
    for _ in range(3):
        assert lib.hmr_gpu_subpel_planes(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(reps):
            assert lib.hmr_gpu_subpel_planes(*args) == 0
        e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    algo = 2 * (sy * ry + 2 * sc * rc) + 16 * sy * ry + 2 * 64 * sc * rc
    lib.hmr_gpu_destroy(ctx)
    gbs = algo / (ms * 1e-3) / 1e9
    return {"kernels": "k_subpel_luma + 2 x k_subpel_chroma (one reference picture)", "bound": "hbm", "algorithmic_bytes_per_picture": int(algo), "ms_per_picture": round(ms, 4),
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "width": width, "height": height}


def multi_stream(lib, device, width, height, keys, streams, warmup, steps):
    """Aggregate throughput of `streams` independent sequences encoded CONCURRENTLY on one GPU (one encoder instance, HIP stream and host thread each).
    One 1080p encode occupies 17 of the 256 CUs (one workgroup per CTU row), so this is what the device delivers when it is kept busy; it is
    reported beside `value`, which stays the single-sequence rate of the BASELINE configuration."""
    import threading
    import encoder_cases as ec
    frames = ec.clip_frames(width, height, warmup + steps)
    gate = threading.Barrier(streams + 1)
    errors, ctu_ms = [], []

    def worker():
        try:
            ctx, enc = C.c_void_p(), C.c_void_p()
            assert lib.hmr_gpu_create(C.byref(ctx), device, None) == 0
            cfg = ec.default_cfg(width, height, **keys)
            assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0
            for f, planes in enumerate(frames):
                assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
            buf, n = C.create_string_buffer(4 << 20), C.c_long()
            for f in range(warmup):
                assert lib.hmr_gpu_enc_encode_source(enc, f, 0, buf, len(buf), C.byref(n), None) in (1, 2)
            gate.wait()
            for f in range(warmup, warmup + steps):
                assert lib.hmr_gpu_enc_encode_source(enc, f, 0, buf, len(buf), C.byref(n), None) in (1, 2)
                p, k, ms, tot = C.c_int(), C.c_int(), C.c_float(), C.c_float()
                lib.hmr_gpu_enc_last_stats(enc, C.byref(p), C.byref(k), C.byref(ms), C.byref(tot))
                ctu_ms.append(ms.value)
            gate.wait()
            lib.hmr_gpu_enc_destroy(enc)
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))
            gate.abort()

    th = [threading.Thread(target=worker) for _ in range(streams)]
    for t in th:
        t.start()
    try:
        gate.wait()
        t0 = time.perf_counter()
        gate.wait()
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = None
    for t in th:
        t.join()
    if dt is None or errors:
        return {"streams": streams, "error": errors[:1]}
    return {"streams": streams, "frames_per_stream": steps, "value": round(streams * steps / dt, 4), "unit": "frames/s aggregate",
            "ctu_stage_ms_per_frame_mean": round(sum(ctu_ms) / max(len(ctu_ms), 1), 1),
            "note": f"{streams} independent {width}x{height} sequences concurrently on one GPU, {steps} timed frames each after {warmup} warm-up frames"}


def multi_stream_batched(lib, device, width, height, keys, streams, warmup, steps):
    """The same measurement through hmr_gpu_enc_encode_batch: one launch per step for the CTU stages of all sequences (no dependence on how streams map
    to hardware queues); each sequence finishes its frame on its own stream and host thread inside the call."""
    import encoder_cases as ec
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    frames = ec.clip_frames(width, height, warmup + steps)
    encs, bufs = [], []
    for _ in range(streams):
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), device, None) == 0
        cfg = ec.default_cfg(width, height, **keys)
        assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(frames):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
        encs.append(enc)
        bufs.append(C.create_string_buffer(4 << 20))
    n = streams
    e_arr = (C.c_void_p * n)(*encs)
    ptrs = (C.c_char_p * n)(*[C.cast(b, C.c_char_p) for b in bufs])
    caps = (C.c_long * n)(*[len(b) for b in bufs])
    got = (C.c_long * n)()
    md5 = [hashlib.md5() for _ in range(n)]
    t0 = None
    for f in range(warmup + steps):
        if f == warmup:
            t0 = time.perf_counter()
        assert lib.hmr_gpu_enc_encode_batch(e_arr, n, (C.c_int * n)(*([f] * n)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        for i in range(n):
            md5[i].update(C.string_at(bufs[i], got[i]))       # (.raw would copy the whole buffer first)
    dt = time.perf_counter() - t0
    p, k, ms, tot = C.c_int(), C.c_int(), C.c_float(), C.c_float()
    lib.hmr_gpu_enc_last_stats(encs[0], C.byref(p), C.byref(k), C.byref(ms), C.byref(tot))
    for enc in encs:
        lib.hmr_gpu_enc_destroy(enc)
    return {"streams": streams, "frames_per_stream": steps, "value": round(streams * steps / dt, 4), "unit": "frames/s aggregate", "ctu_kernel_ms_last_step": round(ms.value, 1),
            "all_streams_identical": len({m.hexdigest() for m in md5}) == 1, "stream_md5": md5[0].hexdigest(),
            "note": f"{streams} independent {width}x{height} sequences, one hmr_gpu_enc_encode_batch call per step ({streams * ((height + 63) // 64)} workgroups in one launch), "
                    f"{steps} timed steps after {warmup}"}


def multi_stream_child(device, workload, streams, batched=False):
    """multi_stream in a fresh process (a child, not an exec): the hardware queues of this process are already shared out among torch's and the
    earlier encoders' streams, and two sequences that land on one queue run one after the other."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--multi-stream-child", str(streams), "--workload", workload, "--device", str(device)] + (["--batched"] if batched else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"streams": streams, "error": (r.stderr or r.stdout)[-300:]}


def measure(step, warmup, nframes, world, device_sync, device, flush=None):
    """The timing contract: `warmup` untimed steps, then steps warmup..nframes-1 bracketed by a barrier + device synchronisation on both sides;
    returns the wall time, MAX over the ranks.  step(f) encodes frame f.  `flush` (pipelined steps: step f delivers the access units of step f - 1) is called
    after the warm-up and again at the end of the timed region, so that the region contains all the work of its steps and nothing of the warm-up's.
    (tests/test_bench_gloo.py runs this with world 2 on gloo.)"""
    import torch
    import torch.distributed as dist

    def fence():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    for f in range(warmup):
        step(f)
    if flush:
        flush()
    fence()
    t0 = time.perf_counter()
    for f in range(warmup, nframes):
        step(f)
    if flush:
        flush()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, 127.0.0.1), rank 0's
    stdout passed through.  Returns the worst exit code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    return max(abs(p.wait()) for p in procs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2-1080p-encode", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=24)
    ap.add_argument("--no-single-thread-order", action="store_true", help="skip the second measurement (wfpp_num_threads = 1)")
    ap.add_argument("--sequences", type=int, default=256, help="independent sequences per GPU, encoded with one launch per step (hmr_gpu_enc_encode_batch: their CTUs are a pool of tasks for "
                    "four row workers per CU); 1 = a single sequence")
    ap.add_argument("--no-pipeline", action="store_true", help="batch steps through hmr_gpu_enc_encode_batch (access units inside the call) instead of the pipelined call")
    ap.add_argument("--streams", type=int, default=0, help="concurrent sequences of the extra multi_stream measurement (0 = skip)")
    ap.add_argument("--multi-stream-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--batched", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--batch", type=int, default=0, help="sequences of the multi_stream_batched measurement (one launch for all CTU stages; 0 = skip)")
    a = ap.parse_args()
    if a.multi_stream_child:
        width, height, keys = WORKLOADS[a.workload]
        fn = multi_stream_batched if a.batched else multi_stream
        print(json.dumps(fn(load_lib(), a.device, width, height, keys, a.multi_stream_child, 2, 3)))
        return

    # --gpus N: under a launcher (torch.distributed.run sets WORLD_SIZE) the flag must agree with it; without one this process starts the N ranks itself - plain child
    # processes, started before anything here has touched the GPU (never an exec of a process that has) - and passes on rank 0's line and the worst exit code
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != a.gpus:
            print(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}", file=sys.stderr)
            sys.exit(2)
    elif a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))

    import torch
    import torch.distributed as dist
    import encoder_cases as ec
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HOMER_BENCH_ONE_DEVICE"):     # test aid: all ranks on GPU 0 (a one-GPU box); RCCL wants a GPU per rank, so the ring then goes over gloo
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("HOMER_BENCH_ONE_DEVICE"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    lib = load_lib()
    if world > 1:
        out = run_engine_ring(a, world, rank, local, torch)
        if rank == 0:
            if not a.no_cpu_baseline:
                width, height, keys = WORKLOADS[a.workload]
                out["cpu_baseline"] = cpu_baseline(width, height, {k: v for k, v in keys.items() if k != "wpp"}, a.cpu_frames)
            emit(out)
        dist.destroy_process_group()
        if not out["stream_matches_reference"]:
            print("bench.py: access units differ from the reference's", file=sys.stderr)
            sys.exit(3)
        return
    out = run_workload(lib, a, a.workload, world, rank, local, torch, sequences=a.sequences)
    if rank == 0:
        width, height, keys = WORKLOADS[a.workload]
        if world == 1 and a.sequences > 1:
            one = run_workload(lib, a, a.workload, world, rank, local, torch)
            out["single_sequence"] = {k: one[k] for k in ("value", "unit", "ms_per_step", "stream_md5", "stream_matches_reference", "frames_checked_against_reference", "schedule")}
            out["single_sequence"]["note"] = "one sequence alone on the GPU (17 of the 256 CUs busy): the latency of the row-parallel CTU chain"
        if world == 1 and a.workload == "cfg2-1080p-encode" and not a.no_single_thread_order:
            other = run_workload(lib, a, "cfg2-1080p-encode-single-thread-order", world, rank, local, torch)
            out["single_thread_order"] = {k: other[k] for k in ("value", "unit", "ms_per_step", "stream_md5", "stream_matches_reference", "frames_checked_against_reference", "schedule")}
            out["single_thread_order"]["note"] = "the same encode with wfpp_num_threads = 1: output identical to the reference's single-thread run (md5 2f0c3447...), which costs guesses, verification and re-encode passes"
            # the metric's other picture size: 2160p, 34 CTU rows on the reference's maximum of 32 WPP threads, I + P and four timed P frames:
            # a batch of 192 sequences (6528 CTU rows for the 1024 workers of the pool: 96 sequences gave 86.5 frames/s, 128 91.9, 192 95.6 - the head and tail of every
            # picture's wavefront leave workers idle, more pictures fill them) and one sequence alone
            import copy
            b = copy.copy(a)
            b.warmup, b.steps = 2, 4
            big = run_workload(lib, b, "cfg2-2160p-encode", world, rank, local, torch, sequences=192 if a.sequences > 1 else 1)
            out["at_2160p"] = {k: big[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "stream_md5", "stream_matches_reference", "frames_checked_against_reference", "clips")}
            out["at_2160p"]["config"] = big["config"]
            if a.sequences > 1:
                big1 = run_workload(lib, b, "cfg2-2160p-encode", world, rank, local, torch)
                out["at_2160p"]["single_sequence"] = {k: big1[k] for k in ("value", "ms_per_step", "stream_matches_reference")}
        if world == 1 and a.workload == "cfg2-1080p-encode" and not a.no_single_thread_order and a.sequences > 1:
            # the reference's deterministic single-thread order (wfpp_num_threads = 1: BASELINE.md's parity mode, md5 2f0c3447... for the published clip, no pinned
            # interleaving) as a BATCH: hmr_gpu_enc_create_serial_pool runs a picture CTU by CTU in raster order as tasks of the pool - one decision in flight per
            # picture, so the launch takes the batch call's maximum of 512 pictures
            import copy
            b = copy.copy(a)
            b.warmup, b.steps = 2, 4
            ser = run_workload(lib, b, "cfg2-1080p-encode-single-thread-order", world, rank, local, torch, sequences=512, serial_pool=True)
            out["serial_order_batch"] = {k: ser[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "stream_md5", "stream_matches_reference", "frames_checked_against_reference", "clips")}
            out["serial_order_batch"]["sequences"] = 512
            out["serial_order_batch"]["note"] = ("wfpp_num_threads = 1, the reference's single-thread order, 512 sequences (eight clips) per launch: every stream equal to the plain "
                                                 "reference's (ref_lockstep, no turnstile); one sequence alone in this order: single_thread_order")
        if world == 1 and a.workload == "cfg2-1080p-encode" and not a.no_single_thread_order:
            # ONE sequence with its engines overlapped on this GPU (hmr_gpu_enc_encode_chain: the frames of a chain in one launch of the CTU kernel, every engine
            # starting its next frame when its last is finished; tools/chain_bench.py): the reference's num_enc_engines pipeline, streams checked against its digests
            import chain_bench
            lanes = []
            for wl, sets in (("cfg2-1080p-encode-engines3", 5), ("cfg2-1080p-encode-engines8", 2), ("cfg2-2160p-encode-engines8", 2)):
                if wl in REFERENCE_MD5:
                    r = chain_bench.run(lib, wl, sets=sets, quiet=True)
                    lanes.append({k: r[k] for k in ("workload", "engines", "objects_per_engine", "chain", "frames", "frames_per_s_full_chains", "frames_per_s_after_first_chain", "stream_matches_reference",
                                                   "ctu_launch_ms_per_chain")})
            out.setdefault("single_sequence", {})["engines_overlapped"] = lanes
            # BASELINE.json configs[2]: 2160p CBR 20000 kbps, performance_mode 1 - a batch of 128 sequences (32: 56 frames/s, 64: 97, 96: 115, 128: 122 - rate control gates a
            # picture's wavefront steps on its entropy coder, so a launch needs many pictures) and one sequence alone (the fixture covers ten frames)
            import copy
            c3 = copy.copy(a)
            c3.warmup, c3.steps = 2, 6
            r3 = run_workload(lib, c3, "cfg3-2160p-cbr", world, rank, local, torch, sequences=128 if a.sequences > 1 else 1)
            out["cfg3_2160p_cbr"] = {k: r3[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "stream_md5", "stream_matches_reference", "frames_checked_against_reference", "clips")}
            out["cfg3_2160p_cbr"]["config"] = dict(r3["config"], bitrate_mode="CBR", bitrate_kbps=20000, vbv_size_kbit=20000, performance_mode=1)
            if a.sequences > 1:
                r31 = run_workload(lib, c3, "cfg3-2160p-cbr", world, rank, local, torch)
                out["cfg3_2160p_cbr"]["single_sequence"] = {k: r31[k] for k in ("value", "ms_per_step", "stream_matches_reference")}
            # BASELINE.json configs[4]: 2160p all-intra, full RDO, intra TU depth 4 - a batch of 96 sequences (32: 10.8 frames/s, 64: 14.8, 96: 15.9) and one sequence alone
            # (the fixture covers eight frames)
            if "cfg5-2160p-intra-rdfull" in REFERENCE_MD5:
                c5 = copy.copy(a)
                c5.warmup, c5.steps = 1, 3
                r5 = run_workload(lib, c5, "cfg5-2160p-intra-rdfull", world, rank, local, torch, sequences=96 if a.sequences > 1 else 1)
                out["cfg5_2160p_intra_rdfull"] = {k: r5[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "stream_md5", "stream_matches_reference", "frames_checked_against_reference", "clips")}
                out["cfg5_2160p_intra_rdfull"]["config"] = r5["config"]
                if a.sequences > 1:
                    r51 = run_workload(lib, c5, "cfg5-2160p-intra-rdfull", world, rank, local, torch)
                    out["cfg5_2160p_intra_rdfull"]["single_sequence"] = {k: r51[k] for k in ("value", "ms_per_step", "stream_matches_reference")}
        if world == 1:
            out["roofline"]["subpel_planes"] = subpel_planes_roofline(lib, torch, width, height)
            probe = valu_issue_probe(lib, local)
            if out["roofline"].get("issue_bound") is not None:
                ib = out["roofline"]["issue_bound"]
                ib["probe"] = probe
                # the ceiling a plain vector instruction meets on THIS device (v_add_u32, two wavefronts per SIMD) instead of the guide's four cycles per instruction,
                # and what ONE wavefront per SIMD reaches (k_encode_pool has 1.5 per SIMD, each a dependent chain most of the time)
                plain = max(r["wave_instructions_per_s"] for r in probe["rows"] if r["instruction"] == "v_add_u32" and not r["dependent_chain"]) * 1e9
                one = [r["wave_instructions_per_s"] for r in probe["rows"] if r["instruction"] == "v_add_u32" and r["waves_per_simd"] == 1 and not r["dependent_chain"]][0] * 1e9
                dep = [r["wave_instructions_per_s"] for r in probe["rows"] if r["instruction"] == "v_add_u32" and r["waves_per_simd"] == 1 and r["dependent_chain"]][0] * 1e9
                fps_kernel = ib["valu_issue_frac"] * ib["valu_issue_peak_per_s"] / ib["valu_wave_instructions_per_frame"]
                ib["guide_four_cycle_peak_per_s"] = ib["valu_issue_peak_per_s"]
                ib["valu_issue_peak_per_s"] = plain
                ib["valu_issue_frac"] = round(ib["valu_wave_instructions_per_frame"] * fps_kernel / plain, 4)
                ib["one_wavefront_per_simd_peak_per_s"] = one
                ib["one_dependent_chain_per_simd_per_s"] = dep
                ib["verdict"] = ("plain vector instructions (v_add_u32, v_mov_b32) issue every 2.5 cycles per SIMD with two wavefronts on it, v_mad_u32_u24 / v_perm_b32 every 5; ONE wavefront gets an "
                                 "instruction every 5 cycles and a dependent chain one every 8.3: at two wavefronts per SIMD (eight per CU: four workers and their helpers) the workers are bound by their own issue latency and the memory "
                                 "trips between instructions, not by the SIMDs' throughput")
            else:
                out["roofline"]["issue_probe"] = probe
        if world == 1 and a.streams > 1:
            out["multi_stream"] = multi_stream_child(local, a.workload, a.streams)
        if world == 1 and a.batch > 1:
            out["multi_stream_batched"] = multi_stream_child(local, a.workload, a.batch, batched=True)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(width, height, {k: v for k, v in keys.items() if k != "wpp"}, a.cpu_frames)
            # the host's figures beside the other sections too: the same three measurements (one process, one thread per CTU row, a process per usable core) of the
            # compiled reference on that section's configuration, on a few frames (2160p: 6; all-intra RD_FULL: 3)
            for section, wl, frames in (("at_2160p", "cfg2-2160p-encode", 6), ("cfg3_2160p_cbr", "cfg3-2160p-cbr", 6), ("cfg5_2160p_intra_rdfull", "cfg5-2160p-intra-rdfull", 3)):
                if section in out:
                    w2, h2, k2 = WORKLOADS[wl]
                    out[section]["cpu_baseline"] = cpu_baseline(w2, h2, {k: v for k, v in k2.items() if k != "wpp"}, frames)
        emit(out)
        # a run whose output differs from the reference's is a failed run, whatever it measured: every section that checked its stream must have matched
        bad = [name for name, sec in [("headline", out)] + [(k, v) for k, v in out.items() if isinstance(v, dict)] if sec.get("stream_matches_reference") is False]
        bad += [f"single_sequence.engines_overlapped[{i}]" for i, lane in enumerate(out.get("single_sequence", {}).get("engines_overlapped", [])) if lane.get("stream_matches_reference") is False]
        if bad:
            print("bench.py: output differs from the reference in: " + ", ".join(bad), file=sys.stderr)
            sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


def run_engine_ring(a, world, rank, local, torch, adapter=None):
    """--gpus N > 1: the engine ring (module docstring).  `adapter`: the encoder behind homerhevc_amd.engines.EngineRing - the product's GpuEngines unless a test brings its
    own (tests/test_bench_gloo.py runs this function at world 2 over gloo with the one-lane checker build behind it, `adapter.on_cpu`)."""
    import torch.distributed as dist
    import encoder_cases as ec
    from homerhevc_amd.engines import EngineRing, GpuEngines
    base = a.workload
    width, height, keys = WORKLOADS[base]
    name = f"{base}-engines{world}"
    keys = dict(keys, engines=world)
    nframes = a.warmup + a.steps
    on_cpu = adapter is not None and getattr(adapter, "on_cpu", False)
    if adapter is None:
        # (HOMER_BENCH_ONE_DEVICE: all ranks on one GPU over gloo - the pictures then cross page-locked host buffers)
        adapter = GpuEngines(lambda seq: ec.default_cfg(width, height, **keys), local, pipelined=not a.no_pipeline, host_exchange=bool(os.environ.get("HOMER_BENCH_ONE_DEVICE")))
    host_tensors = on_cpu or getattr(adapter, "host_exchange", False)
    # Every rank keeps an engine object of EVERY sequence in flight (the engine of that sequence that lives on this rank: its pictures, CTU records, levels) beside the
    # phase planes of the sequences it encodes in a step.  The per-GPU load is the N = 1 line's (--sequences, 256) wherever that fits the GPU's memory; where it does not
    # (eight ranks: 2048 objects) it is what fits, the same on every rank - `sequences_per_gpu` says which.
    per_gpu = a.sequences
    if not on_cpu:
        free0, _ = torch.cuda.mem_get_info()
        probe = adapter.create(0, 0)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        adapter.destroy(probe)
        obj = max(free0 - free1, 1)
        sy, ry, sc, rc = width + 160, height + 160, width // 2 + 80, height // 2 + 80
        planes = 16 * sy * ry + 2 * 64 * sc * rc + 8 * width * height       # phase planes of a picture being encoded + its share of staging
        fit = int(0.85 * free1 // (world * obj + planes))
        if os.environ.get("HOMER_BENCH_ONE_DEVICE"):
            fit //= world
        t = torch.tensor([max(1, min(per_gpu, fit))], dtype=torch.int64, device="cpu" if host_tensors else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        per_gpu = int(t.item())
    S = per_gpu * world
    ring = EngineRing(adapter, S, rank, world)
    ring.load_sources(ec.clip_frames(width, height, nframes))
    gold = REFERENCE_MD5.get(name, {}).get("au_md5", [])
    bad, checked, produced = [0], [0], [0]
    first_bad = []
    pipelined = getattr(adapter, "pipelined", False)
    kept = []
    launch_ms = []

    def step(f):
        units = ring.step(f, last=f + 1 == nframes)                 # (pipelined: the units of frame f - world of the same sequences)
        if f >= a.warmup and getattr(adapter, "last_ctu_ms", None) is not None:
            launch_ms.append(adapter.last_ctu_ms)
        if units:
            kept.append((ring.delivered, units))                    # (hashed after the timed region: checking the output is the harness's work)

    def flush():
        kept.extend(ring.flush())

    dt = measure(step, a.warmup, nframes, world, (lambda: None) if on_cpu else torch.cuda.synchronize, "cpu" if host_tensors else "cuda", flush if pipelined else None)
    for f, units in kept:
        for s, au in units.items():
            produced[0] += 1
            if f < len(gold):
                checked[0] += 1
                if os.environ.get("HOMER_BENCH_DUMP_UNITS") and f in (1, 2, 3, 4):      # (debugging aid: the access units of a few frames, all sequences, every rank)
                    os.makedirs(os.environ["HOMER_BENCH_DUMP_UNITS"], exist_ok=True)
                    open(os.path.join(os.environ["HOMER_BENCH_DUMP_UNITS"], f"au_f{f}_s{s}_{'ok' if hashlib.md5(au).hexdigest() == gold[f] else 'BAD'}.bin"), "wb").write(au)
                if hashlib.md5(au).hexdigest() != gold[f]:
                    bad[0] += 1
                    if len(first_bad) < 6:
                        first_bad.append({"rank": rank, "sequence": s, "frame": f, "bytes": len(au)})
    t = torch.tensor([bad[0], checked[0], produced[0]], dtype=torch.int64, device="cpu" if host_tensors else "cuda")
    dist.all_reduce(t)
    bad_all, checked_all, produced_all = (int(x) for x in t.tolist())
    row_bytes = getattr(adapter, "row_bytes", None)
    matches = bool(checked_all > 0 and bad_all == 0)
    # the dominant kernel on this rank, as on one GPU: k_encode_pool, one launch per step for this rank's `per_gpu` pictures (HIP events on the lead encoder's stream)
    roofline = None
    if launch_ms:
        algo = 10.5 * width * height * per_gpu
        ms = sum(launch_ms) / len(launch_ms)
        roofline = {"bound": "hbm", "kernel": "k_encode_pool", "achieved": round(algo / (ms * 1e-3) / 1e9, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 7),
                    "traffic": None, "launches": len(launch_ms), "algorithmic_bytes_per_launch": int(algo), "ms_per_launch": round(ms, 2), "rank": rank,
                    "note": "rank 0's launches; the kernel and its limits are the N = 1 line's (roofline.issue_bound there)"}
    return {
        "metric": "encoded frames/sec, 1080p & 2160p YUV420 fixed-QP IPPP, 1/2/4/8 MI355X", "value": round(S * a.steps / dt, 4), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int16", "data": "synthetic",
        "config": {"workload": name, "sequences_per_gpu": per_gpu, "sequences_per_gpu_asked": a.sequences, "sequences": S, "frames_per_step": S, "num_enc_engines": world, "wfpp_num_threads": int(keys.get("wpp", 1)),
                   "width": width, "height": height, "frames_in_sequence": nframes, "gop": "IPPP intra_period=100", "qp": 32, "rd_mode": 2, "performance_mode": 2, "sao": 1,
                   "parallelism": f"engine per GPU: frame t of sequence s on rank (s + t) mod {world}; every rank encodes {per_gpu} frames per step in one launch, "
                                  "then one packed RCCL send / recv of the reconstructed pictures (8-bit, without margins) + frame scalars to the next rank",
                   "timed_region": "per step: import of the previous rank's pictures (widen + pad), CTU decisions + filters + SAO + CABAC on the device, headers / NAL on the host, export + ring exchange",
                   "call": "hmr_gpu_enc_encode_batch_pipelined per set of sequences (a rank's sets take turns: a set's access units come with its next call, `world` steps later, "
                           "their download and entropy coding under that call's CTU launch); every pipeline empty when the timed region starts and flushed inside it" if pipelined
                           else "hmr_gpu_enc_encode_batch",
                   "stream_matches_reference": matches, "access_units_checked_against_reference": checked_all},
        "stream_matches_reference": matches, "access_units_checked_against_reference": checked_all, "access_units_produced": produced_all,
        "access_units_differing": bad_all, "first_differences_on_rank_0": first_bad,
        "exchange": {"bytes_per_sequence_and_step": row_bytes, "bytes_per_rank_and_step": row_bytes * per_gpu if row_bytes else None, "collective": "ring of point-to-point transfers (batch_isend_irecv), no reduction"},
        "roofline": roofline, "cpu_baseline": None,
    }


def run_workload(lib, a, workload, world, rank, local, torch, sequences=1, serial_pool=False):
    """`sequences` independent sequences of the workload per GPU: 1 = hmr_gpu_enc_encode_source frame by frame; more = one batch call per step (ONE launch for the
    CTU stages of all of them: a pool of CTU tasks on four row workers per CU; the pipelined call by default, whose access units come with the next call), every
    access unit kept and checked against the reference's digests after the timed region."""
    import encoder_cases as ec
    width, height, keys = WORKLOADS[workload]
    keys = dict(keys)
    image_type = 3 if keys.pop("force_intra", 0) else 0      # (encoder_in_out_t.image_type: IMAGE_I on every picture)
    nframes = a.warmup + a.steps
    # (serial_pool: wfpp_num_threads = 1 as a batch - hmr_gpu_enc_create_serial_pool, the pool's raster schedule)
    S = sequences if int(keys.get("wpp", 1)) > 1 or serial_pool else 1
    # sequence i encodes clip i mod 8 where the reference's digests of that clip cover the run (else the published clip)
    seeds = [sd for sd in CLIP_SEEDS if REFERENCE_MD5.get(seed_workload(workload, sd), {}).get("frames", 0) >= nframes] if S > 1 else []
    if not seeds or os.environ.get("HOMER_BENCH_ONE_CLIP"):
        seeds = [1234]
    clips = {sd: ec.clip_frames(width, height, nframes, seed=sd) for sd in seeds}
    encs, bufs, ctxs = [], [], []
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    for i in range(S):
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), local, None) == 0, lib.hmr_gpu_last_error()
        cfg = ec.default_cfg(width, height, **keys)
        assert (lib.hmr_gpu_enc_create_serial_pool if serial_pool else lib.hmr_gpu_enc_create)(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(clips[seeds[i % len(seeds)]]):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
        encs.append(enc)
        ctxs.append(ctx)
        bufs.append(C.create_string_buffer(4 << 20))
    enc, buf = encs[0], bufs[0]
    nbytes = C.c_long()
    md5s = [hashlib.md5() for _ in range(S)]
    nrep = min(S, len(seeds))      # sequences 0 .. nrep - 1: the first of each clip, whose stream is checked access unit by access unit
    cumulative = [[] for _ in range(nrep)]      # digest of their streams after every access unit
    stats = []
    step_wall = []
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_enc_encode_batch_pipelined.argtypes = lib.hmr_gpu_enc_encode_batch.argtypes
    pipelined = S > 1 and not a.no_pipeline
    e_arr = (C.c_void_p * S)(*encs)
    ptrs = (C.c_char_p * S)(*[C.cast(b, C.c_char_p) for b in bufs])
    caps = (C.c_long * S)(*[len(b) for b in bufs])
    got = (C.c_long * S)()
    kept = []

    def take_units():
        # (pipelined: the access units of the step before; nothing after the first call of a run)
        if pipelined and got[0] == 0:
            return
        # the access units are taken out of the call's buffers here and hashed after the timed region (9 MB of md5 per step in this interpreter would sit between
        # one step's end and the next step's launch: checking the output is the harness's work, not the encoder's)
        kept.append([C.string_at(bufs[i], got[i]) for i in range(S)])

    def flush():
        assert lib.hmr_gpu_enc_encode_batch_pipelined(e_arr, S, None, None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        take_units()

    def step(f):
        t_step = time.perf_counter()
        if S == 1:
            st = lib.hmr_gpu_enc_encode_source(enc, f, image_type, buf, len(buf), C.byref(nbytes), None)
            assert st in (1, 2), lib.hmr_gpu_last_error()
            md5s[0].update(C.string_at(buf, nbytes.value))
        else:
            call = lib.hmr_gpu_enc_encode_batch_pipelined if pipelined else lib.hmr_gpu_enc_encode_batch
            assert call(e_arr, S, (C.c_int * S)(*([f] * S)), (C.c_int * S)(*([image_type] * S)) if image_type else None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
            take_units()
            st, nbytes.value = 0, got[0]
        if S == 1:
            cumulative[0].append(md5s[0].hexdigest())
        p, n, ms, tot = C.c_int(), C.c_int(), C.c_float(), C.c_float()
        lib.hmr_gpu_enc_last_stats(enc, C.byref(p), C.byref(n), C.byref(ms), C.byref(tot))
        stats.append((f, st, nbytes.value, p.value, n.value, ms.value, tot.value))
        step_wall.append(time.perf_counter() - t_step)

    dt = measure(step, a.warmup, nframes, world, torch.cuda.synchronize, "cuda", flush if pipelined else None)
    for units in kept:              # every access unit of every sequence, in the order delivered
        for i in range(S):
            md5s[i].update(units[i])
        for i in range(nrep):
            cumulative[i].append(md5s[i].hexdigest())
    # the one documented exception to byte identity, counted by the library (include/homer_gpu.h: hmr_gpu_enc_stale_predictions): 0 = it did not occur in these clips
    stale_total = 0
    lib.hmr_gpu_enc_stale_predictions.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    for x in encs:
        tot = C.c_long()
        lib.hmr_gpu_enc_stale_predictions(x, None, C.byref(tot))
        stale_total += tot.value
        lib.hmr_gpu_enc_destroy(x)
    for x in ctxs:
        lib.hmr_gpu_destroy(x)          # (a context holds pinned host memory, HBM staging and a stream)
    md5 = md5s[0]
    # every sequence of a clip must have produced the stream of the clip's first sequence, and that one the reference's, access unit by access unit
    all_same = all(md5s[i].hexdigest() == md5s[i % nrep].hexdigest() for i in range(S))

    if True:
        timed = stats[a.warmup:]
        nctu = ((width + 63) // 64) * ((height + 63) // 64)
        launches = sum(s[3] for s in timed)                       # one k_encode_ctus launch per pass
        ctu_ms = sum(s[5] for s in timed)                         # HIP events around the passes of each frame, on the encoder's stream
        frame_ms = sum(s[6] for s in timed)
        # SURVEY.md 8-d, the CTU stage's share of the frame-level compulsory traffic: source + reference + reconstruction (1 byte samples) + levels (2 bytes)
        # ... plus the in-loop filters' read and write of the picture, which are tasks of the same kernel: 10.5 W H bytes per P frame ("whole P frame")
        algo_bytes_frame = 10.5 * width * height
        achieved = S * algo_bytes_frame * len(timed) / (ctu_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed counter passes of this same command: TCC_EA0 requests by their width (64 / 128-byte reads, 64-byte full-line and 32-byte
        # partial-line writes; calibrated on known byte counts, profiles/r04_tcc_calibration.json, as MI355X_MICROARCH.md "HBM" asks for narrow accesses)
        traffic, issue, traffic_source, traffic_build = None, None, None, None
        kernel_name = "k_encode_pool"
        from homerhevc_amd.build import source_digest
        this_build = source_digest()
        pm_path, fm_path = latest_profile("pmc_kernels.json"), latest_profile("traffic_pmc_kernels.json")
        if pm_path and workload == "cfg2-1080p-encode":
            pm = json.load(open(pm_path))
            k, frames_profiled = pm.get(kernel_name), pm.get("frames_encoded_by_k_encode_pool")
            if k and frames_profiled:
                dv = k["derived"]      # (by request width where the passes have it, tools/tcc_calibrate.py; else requests x 64 B)
                if "hbm_read_bytes_TCC_EA0_RDREQ_x64" in dv:
                    per_frame = (dv.get("hbm_read_bytes", dv["hbm_read_bytes_TCC_EA0_RDREQ_x64"]) + dv.get("hbm_write_bytes", dv["hbm_write_bytes_TCC_EA0_WRREQ_x64_upper_bound"])) / frames_profiled
                    traffic = int(per_frame * S * len(timed) / max(launches, 1))       # per launch, like `achieved`
                    traffic_build = pm.get("source_digest") or pm.get("build_commit")
                    traffic_source = (f"{os.path.relpath(pm_path, ROOT)}: TCC_EA0 read / write requests by width, rocprofv3 --pmc passes of this command, bytes per encoded frame x the frames of one launch")
                # the memory-side passes repeated on the final build (tools/pmc_traffic.sh) take precedence for the byte count
                if fm_path:
                    fm = json.load(open(fm_path))
                    fd = fm["k_encode_pool"]["derived"]
                    traffic = int((fd["hbm_read_bytes"] + fd["hbm_write_bytes"]) / fm["frames_encoded_by_k_encode_pool"] * S * len(timed) / max(launches, 1))
                    traffic_build = fm.get("source_digest") or fm.get("build_commit")
                    traffic_source = (f"{os.path.relpath(fm_path, ROOT)}: TCC_EA0 read / write requests by width (64 / 128-byte reads, 64-byte full-line and 32-byte "
                                      "partial-line writes), rocprofv3 --pmc passes of a 256-sequence run of this command, bytes per encoded frame x the frames of one launch")
                # what actually bounds the kernel: wave-instruction issue (MI355X_MICROARCH.md: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction)
                valu_per_frame, salu_per_frame = k["SQ_INSTS_VALU"] / frames_profiled, k["SQ_INSTS_SALU"] / frames_profiled
                fps_kernel = S * len(timed) / (ctu_ms * 1e-3)
                all_insts = sum(k.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_FLAT", "SQ_INSTS_SMEM")) / frames_profiled
                issue = {"valu_wave_instructions_per_frame": int(valu_per_frame), "salu_wave_instructions_per_frame": int(salu_per_frame),
                         "salu_over_valu": round(salu_per_frame / max(valu_per_frame, 1), 3),
                         "valu_issue_peak_per_s": 614.4e9, "valu_issue_frac": round(valu_per_frame * fps_kernel / 614.4e9, 4),
                         "wait_share_of_wave_cycles": k["derived"].get("wait_share_of_wave_cycles"), "issue_share_of_wave_cycles": k["derived"].get("issue_share_of_wave_cycles"),
                         # lanes a vector instruction keeps busy: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) where the passes collected both
                         "valu_lane_utilisation": k["derived"].get("valu_lane_utilisation"),
                         "wavefronts_per_simd": round(2 * WORKERS_PER_CU / 4, 2), "workgroups_per_cu": WORKERS_PER_CU,
                         # a wavefront issues one instruction per four cycles whatever its kind (SQ_ACTIVE_INST_x / SQ_INSTS_x = 1.0 quad-cycles in the counter passes): all
                         # the kernel's wave-instructions x 4 cycles against the cycles its 512 workers had - an upper bound of the workers' issue share (the helpers'
                         # instructions, mailbox polling included, are in the count)
                         "wave_instructions_per_frame": int(all_insts),
                         "issue_cycles_over_worker_cycles": round(4 * all_insts / (256 * WORKERS_PER_CU * 2.4e9 / fps_kernel), 3),
                         "build": pm.get("source_digest") or pm.get("build_commit"),
                         "source": f"{os.path.relpath(pm_path, ROOT)} (rocprofv3 --pmc passes of this command), instruction counts per encoded frame x this run's frames/s of the kernel"}
        digest = md5.hexdigest()
        per_clip = {}
        matches, checked = True, None
        for i in range(nrep):
            ok_i, n_i = check_against_reference(seed_workload(workload, seeds[i]), cumulative[i])
            per_clip[str(seeds[i])] = {"stream_md5": md5s[i].hexdigest(), "matches_reference": bool(ok_i), "frames_checked": n_i, "sequences": len(range(i, S, nrep))}
            matches = matches and ok_i
            checked = n_i if checked is None else min(checked, n_i)
        out = {
            "metric": "encoded frames/sec, 1080p & 2160p YUV420 fixed-QP IPPP, 1/2/4/8 MI355X", "value": round(world * S * a.steps / dt, 4), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": workload, "sequences_per_gpu": S, "frames_per_step": S, "wfpp_num_threads": int(keys.get("wpp", 1)), "width": width, "height": height, "frames_in_sequence": nframes, "gop": "all intra (IMAGE_I forced)" if image_type else "IPPP intra_period=100", "qp": 32,
                       "rd_mode": int(keys.get("rd", 2)), "max_intra_tr_depth": int(keys.get("intra_tr", 2)), "performance_mode": int(keys.get("perf", 2)), "sao": 1, "parallelism": (f"{S} independent sequences per GPU, one launch per step for their CTU stages (a pool of CTU tasks on {WORKERS_PER_CU} row workers per CU)" if S > 1 else "one sequence"),
                       # (the driver keeps `config` verbatim: the run's own verdict on its output sits here too)
                       "stream_matches_reference": bool(matches and all_same), "frames_checked_against_reference": checked,
                       "timed_region": "per frame: phase planes of the reference, CTU decisions, deblocking, SAO statistics / decision / syntax, CABAC of the CTU rows' sub-streams, SAO offsets and border padding on the device "
                                       "(one launch of k_encode_pool: the decisions and the post-decision tasks of enc_post.h), download of the sub-streams, slice header / entry points / NAL escaping on the host; source in HBM; "
                                       "the first warm-up step is the I frames (warmup_ms_per_step[0])",
                       "call": ("hmr_gpu_enc_encode_batch_pipelined: a step's download and entropy coding run under the next step's CTU launch; the pipeline is empty when the timed "
                                "region starts and flushed inside it" if pipelined else "hmr_gpu_enc_encode_batch" if S > 1 else "hmr_gpu_enc_encode_source")},
            "stream_md5": digest, "stream_matches_reference": bool(matches and all_same), "frames_checked_against_reference": checked,
            "evaluations_on_a_stale_prediction_window": stale_total,
            "clips": {"distinct": nrep, "by_seed": per_clip, "sequences_of_a_clip_identical": all_same,
                      "note": "sequence i encodes clip i mod distinct (tools/gen_yuv.py seeds: own texture, pan, pattern, box path); each clip's stream is checked against the compiled reference's digests of that clip"},
            "warmup_ms_per_step": [round(x * 1e3, 1) for x in step_wall[:a.warmup]],
            "schedule": {"ctus_per_frame": nctu, "passes_per_frame": [s[3] for s in timed], "ctu_encodes_per_frame": [s[4] for s in timed],
                         "ctu_stage_ms_per_frame": [round(s[5], 1) for s in timed], "device_ms_per_frame": [round(s[6], 1) for s in timed]},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 7),
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_build": traffic_build, "this_build": this_build,
                         "traffic_is_of_this_build": bool(traffic_build and traffic_build == this_build),
                         "algorithmic_bytes_model": "10.5 W H per frame (SURVEY 8-d whole P frame: CTU stage 7.5 W H + the in-loop filters' read and write)",
                         "achieved_7p5WH_ctu_stage_only": round(achieved * 7.5 / 10.5, 4),
                         "launches": launches, "algorithmic_bytes_per_launch": int(S * algo_bytes_frame * len(timed) / max(launches, 1)), "ms_per_launch": round(ctu_ms / max(launches, 1), 2),
                         "algorithmic_bytes_per_frame": int(algo_bytes_frame), "share_of_device_time": round(ctu_ms / frame_ms, 3),
                         # SURVEY 8-d, the whole P frame (CTU stage + the in-loop filters' read / write): 10.5 W H bytes x frames/s against the same peak
                         "frame_level": {"algorithmic_bytes_per_frame": int(10.5 * width * height), "achieved": round(10.5 * width * height * world * S * a.steps / dt / 1e9, 4),
                                         "frac": round(10.5 * width * height * world * S * a.steps / dt / 1e9 / HBM_PEAK_GBS, 7)},
                         "issue_bound": issue,
                         "note": f"a pool of CTU tasks on {WORKERS_PER_CU} x 256 row workers (one wavefront + one helper each) walking dependent decision chains: bound by the issue "
                                 "latency of the workers' wavefronts (issue_bound) and by memory latency (the workers of an XCD share 4 MB of L2), not by HBM bandwidth; frac is the honest "
                                 "distance from the bandwidth roof"},
            "note": (f"the headline is a BATCH rate: {S} independent sequences share the GPU's {WORKERS_PER_CU * 256} row workers (a 1080p picture has at most 15 CTUs in flight - its wavefront "
                     "steps - so about 120 concurrent sequences are needed to keep the workers busy at all, 256 to keep them busy through the pictures' ramps); ONE sequence alone is "
                     "`single_sequence` (a few frames/s frame by frame, about 10 with its engines overlapped)") if S > 1 else "one sequence alone",
        }
        return out


if __name__ == "__main__":
    main()
