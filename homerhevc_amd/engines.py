"""Frame-parallel encoder engines, one engine per GPU (process).

The reference deals the frames of a sequence to `num_enc_engines` engine threads in decode order (`encoder_engine_thread`,
hmr_encoder_lib.c:3043-3330): frame n goes to engine n mod E, which owns the persistent per-engine state (CTU records, the
WPP threads' contexts), and what the engines share lives in the common `hvenc_enc_t`: the reconstructed reference picture
(`reference_picture_buffer`, filled row by row behind `synchro_sem[1]`, :2393-2445) and a few frame-to-frame scalars
(`avg_dist`, frame typing state; :3185-3279).  With an engine per GPU those two things travel: after frame n, engine n mod E
hands the padded reconstruction and the scalars to engine (n + 1) mod E - a ring of point-to-point transfers, RCCL send / recv
over xGMI (`backend="nccl"`), gloo in the CPU tests.  The stream is the one oracle/ref_ctudump.c's engine turnstile pins on the
reference (include/homer_gpu.h section 12b, enc/enc_host.h).

`EngineRing` runs S sequences at once so that every rank has work at every step: sequence s's frame t is encoded by rank
(s + t) mod E - at step t every rank encodes the frames t of the sequences whose turn it is there (one batch launch on the
GPU), then every rank sends what it produced to the next and receives what the previous produced, in ONE packed transfer.

An `adapter` hides the encoder behind a few calls (the product's is `GpuEngines` below: the C ABI of libhomer_gpu.so on torch
CUDA tensors; the CPU tests bring their own over the checker build):
    create(seq, engine_index) -> handle          load_source(handle, slot, planes)
    encode(handles, slot) -> [access unit bytes] new_buffer(rows) -> tensor [rows, row size] that torch.distributed can send
    export(handle, tensor_row) / import_(handle, tensor_row), or for all engines of a step at once export_many(handles, tensor) / import_many(handles, tensor)

`exchange` (optional) replaces the torch.distributed transfer: exchange(ring, send_rows, recv_rows) - the one-process loop-back of tests/test_gpu_engines.py.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def engine_of(seq, frame, world):
    """the rank (= engine slot of the ring) that encodes `frame` of sequence `seq`"""
    return (seq + frame) % world


def engine_index(seq, rank, world):
    """which of the sequence's E engines lives on `rank`: the one that gets the frames t with (seq + t) mod E == rank"""
    return (rank - seq) % world


class EngineRing:
    def __init__(self, adapter, n_sequences, rank, world, exchange=None):
        self.a, self.S, self.rank, self.world, self.exchange = adapter, n_sequences, rank, world, exchange
        self.enc = {s: adapter.create(s, engine_index(s, rank, world)) for s in range(n_sequences)}
        per_step = (n_sequences + world - 1) // world
        self.send_buf = adapter.new_buffer(per_step)
        self.recv_buf = adapter.new_buffer(per_step)
        self.outstanding = {}       # late-delivering adapters: (sequences of a call) -> the frame whose access units the next call with them returns
        self.delivered = None

    def sequences_at(self, frame, rank=None):
        r = self.rank if rank is None else rank
        return [s for s in range(self.S) if engine_of(s, frame, self.world) == r]

    def load_sources(self, clip):
        """clip[t] = the planes of frame t (the same clip for every sequence): a rank keeps only the frames it will encode"""
        for s, h in self.enc.items():
            for t, planes in enumerate(clip):
                if engine_of(s, t, self.world) == self.rank:
                    self.a.load_source(h, t, planes)

    def step_encode(self, frame, last=False):
        """import what the previous step received, encode frame `frame` of this rank's sequences, export their reconstructions into the send buffer (unless
        `last`).  Returns {sequence: access unit} (see step())."""
        mine = self.sequences_at(frame)
        handles = [self.enc[s] for s in mine]
        if frame > 0 and self.world > 1 and handles:
            if hasattr(self.a, "import_many"):
                self.a.import_many(handles, self.recv_buf)
            else:
                for i, h in enumerate(handles):
                    self.a.import_(h, self.recv_buf[i])
        aus = self.a.encode(handles, frame)
        if getattr(self.a, "pipelined", False):
            key = tuple(mine)
            self.delivered = self.outstanding.get(key)
            self.outstanding[key] = frame
            if self.delivered is None:
                aus = []
        else:
            self.delivered = frame
        if not last and self.world > 1 and handles:
            if hasattr(self.a, "export_many"):
                self.a.export_many(handles, self.send_buf)
            else:
                for i, h in enumerate(handles):
                    self.a.export(h, self.send_buf[i])
        return dict(zip(mine, aus))

    def step_exchange(self, frame):
        """the reconstructions of frame `frame` go to the next rank, the previous rank's arrive: ONE packed transfer each way"""
        if self.world < 2:
            return
        # rank r's sequences of this step are rank r + 1's of the next, in the same order
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        n_out, n_in = len(self.sequences_at(frame)), len(self.sequences_at(frame + 1))
        if self.exchange is not None:
            self.exchange(self, self.send_buf[:n_out], self.recv_buf[:n_in])
            return
        ops = []
        if n_out:
            ops.append(dist.P2POp(dist.isend, self.send_buf[:n_out], nxt))
        if n_in:
            ops.append(dist.P2POp(dist.irecv, self.recv_buf[:n_in], prv))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if os.environ.get("HOMER_RING_TRACE") and not self.recv_buf.is_cuda:      # debugging aid: checksums of what left and what arrived (host buffers only)
            import zlib
            with open(os.path.join(os.environ["HOMER_RING_TRACE"], f"ring_rank{self.rank}.txt"), "a") as f:
                for i in range(n_out):
                    row = self.send_buf[i].numpy().tobytes()
                    rb = self.a.ref_bytes
                    y, c = rb * 2 // 3, rb // 6
                    f.write(f"sent frame {frame} row {i} to {nxt} crc {zlib.crc32(row):08x} y {zlib.crc32(row[:y]):08x} u {zlib.crc32(row[y:y + c]):08x} v {zlib.crc32(row[y + c:rb]):08x} "
                            f"state {row[rb:rb + self.a.state_bytes].hex()[:400]}\n")
                for i in range(n_in):
                    f.write(f"recv frame {frame} row {i} from {prv} crc {zlib.crc32(self.recv_buf[i].numpy().tobytes()):08x}\n")
        # r.wait() orders torch's current stream behind the transfer, not the host: the library reads / writes these buffers on streams of its own
        if self.recv_buf.is_cuda:
            torch.cuda.current_stream(self.recv_buf.device).synchronize()

    def step(self, frame, last=False):
        """encode frame `frame` of this rank's sequences; returns {sequence: access unit}.  Unless `last`, the reconstructions go round the ring afterwards.
        An adapter that delivers late (`GpuEngines(pipelined=True)`: a call returns the access units of the same sequences' previous call, `world` frames earlier,
        whose download and entropy coding ran under this call's CTU launch) makes this {sequence: unit of frame - world} ({} for the first `world` frames); `delivered`
        says which frame the units belong to, `flush()` fetches what is outstanding."""
        out = self.step_encode(frame, last)
        if not last:
            self.step_exchange(frame)
        return out

    def flush(self):
        """late-delivering adapters: [(frame, {sequence: access unit})] for everything outstanding"""
        out = []
        for key, frame in sorted(self.outstanding.items(), key=lambda kv: kv[1]):
            aus = self.a.encode([self.enc[s] for s in key], None)
            out.append((frame, dict(zip(key, aus))))
        self.outstanding = {}
        return out


class GpuEngines:
    """adapter over libhomer_gpu.so (no fallback: without the HIP library there is nothing to run)"""

    def __init__(self, cfg_of, device, pipelined=False, host_exchange=False):
        torch.cuda.init()          # torch's HIP runtime first: it ships its own libamdhip64 and finds no GPU once the system one has been initialised by the library
        self.lib = C.CDLL(os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))
        self.cfg_of, self.device, self.pipelined, self.host_exchange = cfg_of, device, pipelined, host_exchange
        self.scratch = {}
        lib = self.lib
        lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        lib.hmr_gpu_enc_create_engine.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
        lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.hmr_gpu_enc_encode_batch_pipelined.argtypes = lib.hmr_gpu_enc_encode_batch.argtypes
        lib.hmr_gpu_enc_reference_bytes.restype = C.c_long
        lib.hmr_gpu_enc_reference_bytes.argtypes = [C.c_void_p]
        lib.hmr_gpu_enc_export_references8.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_long, C.c_void_p]
        lib.hmr_gpu_enc_import_references8.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_long, C.c_char_p]
        lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
        lib.hmr_gpu_enc_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.hmr_gpu_last_error.restype = C.c_char_p
        self.last_ctu_ms = None
        self.state_bytes = lib.hmr_gpu_enc_state_bytes()
        self.ref_bytes = None
        self.bufs = {}
        self.states = {}
        self.slots = {}

    def create(self, seq, index):
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert self.lib.hmr_gpu_create(C.byref(ctx), self.device, None) == 0, self.lib.hmr_gpu_last_error()
        cfg = self.cfg_of(seq)
        assert self.lib.hmr_gpu_enc_create_engine(ctx, C.byref(cfg), index, C.byref(enc)) == 0, self.lib.hmr_gpu_last_error()
        if self.ref_bytes is None:
            self.ref_bytes = self.lib.hmr_gpu_enc_reference_bytes(enc)
        self.bufs[enc.value] = C.create_string_buffer(4 << 20)
        self.slots[enc.value] = {}
        return enc

    @property
    def row_bytes(self):
        """a picture as it travels (8-bit samples, no margins) + the frame-to-frame scalars, rounded up to 16 bytes"""
        return (self.ref_bytes + self.state_bytes + 15) // 16 * 16

    def new_buffer(self, rows):
        """what torch.distributed sends / receives: on the GPU (RCCL), or page-locked on the host (`host_exchange`: gloo, e.g. several ranks on one GPU)"""
        shape = (max(rows, 1), self.row_bytes)
        if self.host_exchange:
            return torch.zeros(shape, dtype=torch.uint8).pin_memory()
        return self._zeros_on_device(shape)

    def _zeros_on_device(self, shape):
        """a zeroed device buffer the LIBRARY will write through its raw pointer: torch fills it on its own stream, which the library's (non-blocking) streams do
        not wait for - the fill has to be over before the pointer is handed out, or it can land on top of the first pictures written there"""
        t = torch.zeros(shape, dtype=torch.uint8, device=f"cuda:{self.device}")
        torch.cuda.current_stream(t.device).synchronize()
        return t

    def _device_rows(self, buf):
        if buf.is_cuda:
            return buf
        key = tuple(buf.shape)
        if key not in self.scratch:
            self.scratch[key] = self._zeros_on_device(key)
        return self.scratch[key]

    def load_source(self, h, frame, planes):
        slot = len(self.slots[h.value])
        self.slots[h.value][frame] = slot
        assert self.lib.hmr_gpu_enc_load_source(h, slot, *planes) == 0, self.lib.hmr_gpu_last_error()

    def encode(self, handles, frame):
        n = len(handles)
        if n == 0:
            return []
        e_arr = (C.c_void_p * n)(*handles)
        slots = None if frame is None else (C.c_int * n)(*[self.slots[h.value][frame] for h in handles])      # (None: a pipelined flush)
        ptrs = (C.c_char_p * n)(*[C.cast(self.bufs[h.value], C.c_char_p) for h in handles])
        caps = (C.c_long * n)(*[len(self.bufs[h.value]) for h in handles])
        got = (C.c_long * n)()
        call = self.lib.hmr_gpu_enc_encode_batch_pipelined if self.pipelined else self.lib.hmr_gpu_enc_encode_batch
        assert call(e_arr, n, slots, None, ptrs, caps, got) == 0, self.lib.hmr_gpu_last_error()
        if frame is not None:
            # the CTU launch of this call, by HIP events on the lead encoder's stream (what bench.py's roofline divides by)
            p, k, ms, tot = C.c_int(), C.c_int(), C.c_float(), C.c_float()
            self.lib.hmr_gpu_enc_last_stats(handles[0], C.byref(p), C.byref(k), C.byref(ms), C.byref(tot))
            self.last_ctu_ms = ms.value
        return [C.string_at(self.bufs[h.value], got[i]) for i, h in enumerate(handles)]

    def export_many(self, handles, buf):
        """the reconstructions (8-bit, unpadded) and the scalars of the engines of a step into buf[:n]: one call, one copy of all the scalars"""
        n = len(handles)
        dev = self._device_rows(buf)
        states = C.create_string_buffer(self.state_bytes * n)
        e_arr = (C.c_void_p * n)(*handles)
        assert self.lib.hmr_gpu_enc_export_references8(e_arr, n, dev.data_ptr(), self.row_bytes, states) == 0, self.lib.hmr_gpu_last_error()
        tail = torch.frombuffer(bytearray(states.raw), dtype=torch.uint8).view(n, self.state_bytes)
        dev[:n, self.ref_bytes:self.ref_bytes + self.state_bytes].copy_(tail)
        if dev is not buf:
            buf[:n].copy_(dev[:n])
        torch.cuda.current_stream(dev.device).synchronize()

    def import_many(self, handles, buf):
        n = len(handles)
        dev = self._device_rows(buf)
        if dev is not buf:
            dev[:n].copy_(buf[:n])
        states = dev[:n, self.ref_bytes:self.ref_bytes + self.state_bytes].cpu().contiguous().numpy().tobytes()      # (one copy for all engines; it also waits for the rows)
        e_arr = (C.c_void_p * n)(*handles)
        assert self.lib.hmr_gpu_enc_import_references8(e_arr, n, dev.data_ptr(), self.row_bytes, states) == 0, self.lib.hmr_gpu_last_error()

    def destroy(self, h):
        self.bufs.pop(h.value, None)
        self.slots.pop(h.value, None)
        self.lib.hmr_gpu_enc_destroy(h)
