#!/usr/bin/env python3
"""Latency vs throughput probe for the fused kernels: time one batched launch at several job counts (C ABI + HIP events)."""
import ctypes as C
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # gpu_abi.py: the ctypes mirror of the batched ABI
import bench_callmix as bench
from gpu_abi import Context
import torch

def main():
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = Context(device=0, stream=stream.cuda_stream)
    rng = np.random.default_rng(1)
    arena = bench.Arena()
    groups, _ = bench.build_groups(bench.load_callmix(2), rng, arena, fused=True)
    host = np.zeros(arena.size, np.int16)
    for off, data in arena.init:
        host[off:off + data.size] = data
    d_arena = torch.from_numpy(host).to(dev)
    base = C.c_void_p(d_arena.data_ptr())
    for g in groups:
        if g["name"] not in ("me_subpel", "intra_search", "mc_luma", "tu_chain"):
            continue
        jobs = g["jobs"]
        if g["name"] == "tu_chain" and os.environ.get("NO_SBH"):
            jobs = jobs.copy(); jobs["p0"] &= ~np.uint32(1 << 6)
        d_jobs = torch.from_numpy(jobs.view(np.uint8)).to(dev)
        d_out = torch.zeros(8 * len(jobs), dtype=torch.int32, device=dev)
        d_out2 = torch.zeros(len(jobs), dtype=torch.int32, device=dev)
        row = []
        for n in (64, 256, 1024, 4096, len(jobs)):
            n = min(n, len(jobs))
            def launch():
                jp, op = C.c_void_p(d_jobs.data_ptr()), C.c_void_p(d_out.data_ptr())
                if g["name"] == "me_subpel":
                    ctx.call("hmr_gpu_motion_estimation_batch", jp, n, g["size"], base, base, 128, 64, bench.W, bench.HA, op)
                elif g["name"] == "intra_search":
                    ctx.call("hmr_gpu_intra_search_batch", jp, n, g["size"], base, base, base, op)
                elif g["name"] == "mc_luma":
                    ctx.call("hmr_gpu_mc_batch", jp, n, g["size"], 0, base, base)
                else:
                    ctx.call("hmr_gpu_tu_chain_batch", jp, n, g["size"], base, base, base, base, op, C.c_void_p(d_out2.data_ptr()))
            with torch.cuda.stream(stream):
                for _ in range(3):
                    launch()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(20):
                    launch()
                e1.record(stream)
                torch.cuda.synchronize()
            row.append((n, round(e0.elapsed_time(e1) / 20 * 1e3, 1)))
        print(g["name"], g["size"], row, "(jobs, us per launch)")

main()
