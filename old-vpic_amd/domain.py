"""Brick-decomposed domains: one process per GPU, exchanges over torch.distributed (RCCL on GPUs).

What the reference does with one MPI rank per brick (src/grid/partition.c:35-85, ranks ordered
ix + gpx*(iy + gpy*iz)) and its per-step exchanges:
  * tangential-B ghost planes before advance_e      src/field_advance/standard/remote.c:61-134
  * synchronize_jf, three ordered passes x, y, z    src/field_advance/standard/remote.c:416-506
  * boundary_p, num_comm_round = 3 rounds           src/species_advance/standard/boundary_p.c:341-497,
                                                    src/vpic/advance.cxx:94-96, src/vpic/vpic.cxx:17
The box is cut into gpx x gpy x gpz bricks (deck key `topology`; default: x-slabs, gpx = world, what bench.py runs).  An
axis that is cut has a neighbour on either face (a periodic ring along it); an axis that is not wraps onto the domain
itself.  Messages are device buffers packed and unpacked by the engine's kernels; the host only moves counts.  The data
path has no collective: each exchange is a pair of point-to-point transfers per neighbour (xGMI is point-to-point).

The class drives anything with the Engine interface (engine.py); the CPU tests plug the oracle in
through that same interface to check the exchange choreography with the gloo backend.
"""
import contextlib
import time

import numpy as np
import torch
import torch.distributed as dist

from . import layout as L
from .engine import Engine, make_grid

NUM_COMM_ROUND = 3          # src/vpic/vpic.cxx:17


class _HostEvent:
    """Stand-ins for a HIP event / stream where there is no device (CPU rehearsal of the RCCL transport's call
    sequence, deck key comm_stream_rehearsal): they only keep count of what was recorded and waited for."""
    def __init__(self):
        self.stream = None

    def record(self, stream):
        self.stream = stream
        stream.recorded += 1


class _HostStream:
    def __init__(self):
        self.recorded = 0
        self.waited = 0

    def wait_event(self, ev):
        assert ev.stream is not None, "an event is waited for before it was recorded"
        self.waited += 1


class LoopbackTransport:
    """Several domains in ONE process (a thread each) on ONE GPU: a message is a device-to-device copy on the receiver's
    communication stream, ordered behind the sender's packing by an event -- the RCCL transport's stream choreography
    without RCCL (which refuses two ranks on one device).  For traces of what overlaps with what (tools/overlap_trace.py);
    nothing in the product path uses it."""

    def __init__(self):
        import collections
        import queue
        import threading
        self.box = collections.defaultdict(queue.Queue)      # (destination rank, direction) -> (tensor, ready event, sender)
        self.free = collections.defaultdict(lambda: threading.Semaphore(1))   # (sender, direction): its buffer may be packed again
        self.copied = {}                                      # (sender, direction) -> event: the receiver's copy of the previous message

    def move(self, dom, send, recv):
        """Called inside the domain's communication-stream scope, after that stream was told to wait for the packing."""
        comm = torch.cuda.current_stream()
        for d, t in send.items():
            ready = torch.cuda.Event()
            ready.record(comm)
            self.box[(dom._to(d), d)].put((t, ready, dom.rank))
        for d, t in recv.items():
            src, ready, sender = self.box[(dom.rank, d)].get(timeout=120)
            comm.wait_event(ready)
            t.copy_(src[:t.numel()], non_blocking=True)
            done = torch.cuda.Event()
            done.record(comm)
            self.copied[(sender, d)] = done
            self.free[(sender, d, src.data_ptr())].release()

    def before_pack(self, dom, tensors):
        """The buffers are reused every step: wait (host, then the engine's stream) until the receiver has copied the
        previous message out of each."""
        for d, t in tensors.items():
            self.free[(dom.rank, d, t.data_ptr())].acquire(timeout=120)
            ev = self.copied.pop((dom.rank, d), None)
            if ev is not None:
                dom.estream.wait_event(ev)

    def allsum(self, vals):
        raise NotImplementedError("the loopback transport serves the step's point-to-point messages only")


class SlabDomain:
    def __init__(self, deck, rank, world, local_rank=0, engine_factory=None, load=True, push_mode="exact", loopback=None):
        """deck: dict(gx, gy, gz, ppc, dt, q, drift, vth, sort_interval) -- see bench.py.
        loopback: a LoopbackTransport shared by several domains of ONE process (one thread each, all on one GPU): the
        stream choreography of the RCCL transport with device-to-device copies doing the moving (tools/overlap_trace.py)."""
        self.loopback = loopback
        self.rank, self.world, self.deck = rank, world, deck
        gx, gy, gz = deck["gx"], deck["gy"], deck["gz"]
        self.gp = gp = tuple(deck.get("topology", (world, 1, 1)))
        assert gp[0] * gp[1] * gp[2] == world and gx % gp[0] == 0 and gy % gp[1] == 0 and gz % gp[2] == 0
        self.nx, self.ny, self.nz = gx // gp[0], gy // gp[1], gz // gp[2]
        self.coord = (rank % gp[0], (rank // gp[0]) % gp[1], rank // (gp[0] * gp[1]))        # partition.c:41-45
        stride = (1, gp[0], gp[0] * gp[1])
        # the rank behind every face (0..2: low x, y, z; 3..5: high); an axis that is not cut wraps onto this domain
        self.face_rank = [rank] * 6
        for a in range(3):
            i = self.coord[a]
            self.face_rank[a] = rank + ((i - 1) % gp[a] - i) * stride[a]
            self.face_rank[a + 3] = rank + ((i + 1) % gp[a] - i) * stride[a]
        # deck key "self_send": axes that are NOT cut but are to be treated as if they were -- both faces shared with this same
        # domain, every message sent to self (what the reference does on a periodic rank of its own, grid_comm.c:17-19,49).
        # For tests: the whole multi-domain path, RCCL included, on one GPU (needs the vpic_hip_comm transport: torch refuses
        # point-to-point messages to self)
        self.self_send = [a for a in deck.get("self_send", []) if gp[a] == 1]
        cut = [gp[a] > 1 or a in self.self_send for a in range(3)]
        self.axes = [a for a in range(3) if cut[a]]                  # the axes with neighbours
        self.dirs = [d for d in range(6) if cut[d % 3]]              # ... and their faces, in message order
        self.left, self.right = self.face_rank[0], self.face_rank[3]
        fbc = list(self.face_rank)
        for a in self.self_send:                                     # to the engine a face is shared when its code names ANOTHER domain
            fbc[a] = fbc[a + 3] = world + rank
        pbc = list(fbc)
        # walls on axes that are not cut (deck key "walls": {axis: (field code, particle code)}, e.g. the conducting, reflecting
        # z walls of the reconnection decks, turbulence.cxx:265-269): both faces of the axis
        for a, (fcode, pcode) in deck.get("walls", {}).items():
            assert gp[a] == 1, "walls on a cut axis are not served"
            fbc[a] = fbc[a + 3] = fcode
            pbc[a] = pbc[a + 3] = pcode
        # cell size from the GLOBAL box, as partition_periodic_box does (partition.c:60-66)
        g = make_grid(self.nx, self.ny, self.nz, float(gx) / gp[0], float(gy) / gp[1], float(gz) / gp[2], deck["dt"],
                      cvac=deck.get("cvac", 1.0), eps0=deck.get("eps0", 1.0), damp=deck.get("damp", 0.0),
                      fbc=fbc, pbc=pbc, rank=rank)
        self.grid = g
        self.engine = (engine_factory or (lambda grid: Engine(grid, local_rank)))(g)
        e = self.engine
        self.dev = torch.device("cuda", local_rank) if e.device_type == "cuda" else torch.device("cpu")
        e.set_vacuum()
        if push_mode != "exact":
            e.set_push_mode(push_mode)
        if hasattr(e, "set_sort_order"):
            e.set_sort_order("engine")                     # tile order (include/vpic_hip.h): nothing here reads partition[]
        if deck.get("accumulation") and hasattr(e, "set_accumulation"):
            e.set_accumulation(deck["accumulation"], abs(float(deck["q"])))   # "deterministic": 64-bit fixed-point sums
        self.n_per_species = self.nx * self.ny * self.nz * deck["ppc"]
        self.species = []
        if load and "species4" in deck:
            # species that differ (bench.py's reconnection decks): (q/m, sign of the macro-charge or 0 for a tracer copy, drift, thermal spread)
            for k, (q_m, sgn, u, vth) in enumerate(deck["species4"]):
                sp = e.new_species(q_m, int(self.n_per_species * 1.25) + 4096, max(self.n_per_species // 8, 4096))
                e.load_maxwellian(sp, deck["ppc"], 1 + k + 16 * rank, sgn * abs(deck["q"]), u, vth)
                self.species.append(sp)
            e.load_interpolator()
        elif load:
            for k, u in enumerate(deck.get("species", [(deck["drift"], 0.0, 0.0), (-deck["drift"], 0.0, 0.0)])):
                # head room for density fluctuations between slabs
                sp = e.new_species(-1.0, int(self.n_per_species * 1.25) + 4096, max(self.n_per_species // 8, 4096))
                e.load_maxwellian(sp, deck["ppc"], 1 + k + 16 * rank, deck["q"], u, deck["vth"])
                self.species.append(sp)
            e.load_interpolator()
        self.fbuf = {(kind, d): torch.empty(e.face_count(d), dtype=torch.float32, device=self.dev)
                     for kind in ("send", "recv") for d in self.dirs}
        self.cnt_send = {d: torch.zeros(1, dtype=torch.int32, device=self.dev) for d in self.dirs}
        self.cnt_recv = {d: torch.zeros(1, dtype=torch.int32, device=self.dev) for d in self.dirs}
        self.inj_cap = 0
        self.inj = {}
        # which protocol: engines that keep the exchange's counts on the device (the HIP engine) run the
        # device-resident one (one host synchronisation per step, exchanges overlapped with field kernels on a
        # communication stream when the transport is RCCL); anything else (the CPU oracle behind the Engine
        # interface, in the gloo tests) runs the reference's count-then-payload protocol
        self.resident = hasattr(e, "exchange_pack") and not deck.get("legacy_exchange", False)
        self.n_sync = 0                                      # host synchronisations the protocol needs (cumulative)
        self.n_sync_transport = 0
        self.n_step = 0
        if self.resident:
            # capacity (injectors) of the message of one species across each shared face; both ends derive the next
            # step's from the header of this step's (see _next_cap), starting from an eighth of a boundary plane
            dims = (self.nx, self.ny, self.nz)
            self.plane = {d: dims[(d + 1) % 3] * dims[(d + 2) % 3] * deck["ppc"] for d in self.dirs}
            self.cap = {}                                    # (kind, direction, species) -> capacity, filled on first use
            self.cap2 = {}                                   # later rounds (stragglers only): (kind, direction) -> capacity, 4096 to begin with
            self.msg = {}
            self.mover_cap = None                            # first step: the species' full mover capacity
            self.fbuf2 = {(kind, d): torch.empty(e.face_count(d), dtype=torch.float32, device=self.dev)
                          for kind in ("send", "recv") for d in self.dirs}   # tang-B while the jf buffers are in flight
        self.comm = None
        self.group = None
        self.vcomm = None            # the C ABI's RCCL transport (vpic_hip_comm_*, csrc/transport.hip): what the C++ deck host uses too
        # Transport.  RCCL (backend "nccl") moves the device buffers as they are, on a communication stream.  gloo moves
        # host memory only: with HIP engines it is a REHEARSAL transport that has to be asked for (bench.py --backend gloo,
        # the one-GPU tests) and stages every message through the host.  There is no fallback from one to the other: a
        # transport that cannot move device buffers fails here, with the reason, before the first step.
        backend = ("loopback" if loopback is not None else dist.get_backend()) if world > 1 else None
        if self.dev.type == "cuda" and hasattr(e, "_h") and loopback is None and deck.get("transport", "vpic") == "vpic" \
                and (backend == "nccl" or (world == 1 and self.self_send)):
            self._open_vcomm(world, rank)
        if world == 1 and self.self_send and self.vcomm is None:
            raise RuntimeError("self-sends need the vpic_hip_comm transport (a HIP engine): " + getattr(self, "vcomm_error", "not available"))
        self.staged = self.dev.type == "cuda" and world > 1 and backend == "gloo"
        self.transport = ("rccl (vpic_hip_comm: ncclSend / ncclRecv groups on the library's communication stream)" if self.vcomm is not None else "none") \
            if (world == 1 or self.vcomm is not None) else ("gloo (host-staged rehearsal)" if self.staged else "gloo (host tensors)" if self.dev.type == "cpu" else
                                                     "loopback (one process, device-to-device copies)" if loopback is not None else "rccl")
        if getattr(self, "vcomm_error", None) and world > 1:
            self.transport = "rccl through torch.distributed (the library's own transport could not be opened: %s)" % self.vcomm_error
        if self.vcomm is not None:
            self._exchange({d: self.cnt_send[d] for d in self.dirs}, {d: self.cnt_recv[d] for d in self.dirs})   # first contact
            e.sync()
        elif self.dev.type == "cuda" and not self.staged and world > 1 and loopback is None:
            try:                                             # first contact with the transport: a tiny exchange over every shared face
                self._exchange({d: self.cnt_send[d] for d in self.dirs}, {d: self.cnt_recv[d] for d in self.dirs})
                torch.cuda.synchronize(self.dev)
            except Exception as exc:                         # noqa: BLE001 -- whatever the transport raised
                raise RuntimeError(f"[rank {rank}] the {backend} transport cannot exchange device buffers with ranks "
                                   f"{sorted(set(self.face_rank[d] for d in self.dirs))}: {type(exc).__name__}: {exc}") from exc
        if self.vcomm is not None:
            pass                                             # (the library's transport has its own communication stream and events)
        elif self.dev.type == "cuda" and not self.staged and world > 1:
            # RCCL transport: exchanges are enqueued on a (high-priority) communication stream and ordered against the
            # engine's stream with events; the host never waits for them
            self.comm = torch.cuda.Stream(device=self.dev, priority=-1)
            self.estream = torch.cuda.ExternalStream(e.stream(), device=self.dev)
        elif self.dev.type == "cpu" and world > 1 and deck.get("comm_stream_rehearsal", False):
            # CPU rehearsal: the un-staged branch of _start (device tensors handed to the backend as they are)
            self.comm, self.estream = _HostStream(), _HostStream()
        elif self.staged and self.resident and deck.get("comm_stream_rehearsal", False):
            # one-GPU rehearsal of the RCCL path's stream plumbing (external stream, events, stream waits) with the
            # staged gloo transport doing the moving inside the communication stream's scope: the host blocks
            # there, so this exercises the ordering calls, not the overlap
            self.comm = torch.cuda.Stream(device=self.dev)
            self.estream = torch.cuda.ExternalStream(e.stream(), device=self.dev)
        # timing of the exchanges (bench.py switches it on for the timed region: trace_reset / trace_report)
        self.trace = False
        self._tr_pairs, self._tr_waits = [], []
        self._tr_host = dict(step=0.0, blocked=0.0, staged=0.0, steps=0)
        self.n_recovery = 0                                  # extra rounds because a message was full
        self.n_reserved = 0                                  # times a species' arrays were enlarged

    def _open_vcomm(self, world, rank):
        """The C ABI's RCCL transport for this domain's engine: rank 0 makes the id, the launcher's process group hands it round.
        All ranks succeed or all go on with torch.distributed (reported in `transport`)."""
        import ctypes as C
        l = self.engine._l
        idb = (C.c_char * 128)()
        err = None
        try:
            if rank == 0 and l.vpic_hip_comm_unique_id(idb):
                err = l.vpic_hip_last_error().decode()
            if world > 1:
                box = [bytes(idb) if err is None else None]
                dist.broadcast_object_list(box, src=0)
                if box[0] is None:
                    err = err or "rank 0 could not make a communicator id"
                else:
                    idb = (C.c_char * 128).from_buffer_copy(box[0])
            h = C.c_void_p()
            if err is None and l.vpic_hip_comm_create(C.byref(h), self.engine._h, idb, world, rank):
                err = l.vpic_hip_last_error().decode()
        except Exception as exc:                             # noqa: BLE001 -- an older library, a launcher without object broadcast ...
            err = "%s: %s" % (type(exc).__name__, exc)
        if world > 1:                                        # everybody or nobody
            flags = [None] * world
            dist.all_gather_object(flags, err)
            bad = [f for f in flags if f]
            if bad and err is None:
                l.vpic_hip_comm_destroy(h)
                err = "another rank: " + bad[0]
            elif bad:
                err = bad[0] if err is None else err
        if err is None:
            self.vcomm = h
        else:
            self.vcomm_error = err

    def _vstart(self, send, recv):
        import ctypes as C
        ds, dr = [d for d in range(6) if d in send], [d for d in range(6) if d in recv]
        ns, nr = len(ds), len(dr)
        sb = (C.c_void_p * max(ns, 1))(*[send[d].data_ptr() for d in ds])
        sn = (C.c_size_t * max(ns, 1))(*[send[d].numel() * send[d].element_size() for d in ds])
        sp = (C.c_int * max(ns, 1))(*[self._to(d) for d in ds])
        rb = (C.c_void_p * max(nr, 1))(*[recv[d].data_ptr() for d in dr])
        rn = (C.c_size_t * max(nr, 1))(*[recv[d].numel() * recv[d].element_size() for d in dr])
        rp = (C.c_int * max(nr, 1))(*[self._from(d) for d in dr])
        tok = C.c_int(-1)
        if self.engine._l.vpic_hip_comm_start(self.vcomm, ns, sb, sn, sp, nr, rb, rn, rp, C.byref(tok)):
            raise RuntimeError(self.engine._l.vpic_hip_last_error().decode())
        return ("vpic", tok.value)

    def host_syncs_per_step(self):
        return self.n_sync / self.n_step if self.n_step else None

    def _round_cap(self, n):
        g = self.deck.get("exchange_cap_granule", 4096)      # (tests lower it to make messages overflow)
        return max(g, (int(n) + g - 1) // g * g)

    def _next_cap(self, cap, wanted):
        """Capacity of a directed message for the next step, from what its sender wanted to send this step; both
        ends evaluate this on the same two numbers (the receiver reads `wanted` in the header)."""
        return self._round_cap(1.5 * wanted + self.deck.get("exchange_cap_granule", 4096))   # the count changes by a few per cent from one step to the next

    # ---- transport that does not stall the host (RCCL) or does (gloo, staged or CPU) -----------------------------
    def _start(self, send, recv):
        """Post the exchange {direction: tensor}; returns a token for _finish.  With RCCL the transfers are enqueued
        on the communication stream behind everything the engine's stream has been given so far."""
        if self.vcomm is not None:
            return self._vstart(send, recv)
        if self.comm is None:
            t0 = time.perf_counter()
            self._exchange(send, recv)
            if self.trace:
                self._tr_host["staged"] += time.perf_counter() - t0
            return None
        on_device = self.dev.type == "cuda"
        timed = self.trace and on_device
        ev = torch.cuda.Event() if on_device else _HostEvent()
        ev.record(self.estream)
        self.comm.wait_event(ev)
        dev_recv = recv
        with (torch.cuda.stream(self.comm) if on_device else contextlib.nullcontext()):
            if timed:
                t_a = torch.cuda.Event(enable_timing=True)
                t_a.record(self.comm)
            if self.staged:                                  # rehearsal: see __init__
                send = {d: t.cpu() for d, t in send.items()}
                recv = {d: torch.empty_like(t, device="cpu") for d, t in dev_recv.items()}
                self.n_sync_transport += 1
            if self.loopback is not None:
                self.loopback.move(self, send, recv)
                ops = []
            else:
                ops = [dist.P2POp(dist.isend, send[d], self._to(d), group=self.group) for d in range(6) if d in send]
                ops += [dist.P2POp(dist.irecv, recv[d], self._from(d), group=self.group) for d in range(6) if d in recv]
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()                                     # stream-level for RCCL: orders self.comm, not the host
            if self.staged:
                for d, t in recv.items():
                    dev_recv[d].copy_(t)
            done = (torch.cuda.Event(enable_timing=timed) if on_device else _HostEvent())
            done.record(self.comm)
            if timed:
                self._tr_pairs.append((t_a, done))
        return done

    def _finish(self, token):
        """What the engine's stream is given next waits for the exchange."""
        if isinstance(token, tuple):                         # the library's transport
            if self.engine._l.vpic_hip_comm_finish(self.vcomm, token[1]):
                raise RuntimeError(self.engine._l.vpic_hip_last_error().decode())
            return
        if token is not None:
            if self.trace and self.dev.type == "cuda":       # how long the engine's stream stood still for this message
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(self.estream)
                self.estream.wait_event(token)
                b.record(self.estream)
                self._tr_waits.append((a, b))
            else:
                self.estream.wait_event(token)

    def trace_reset(self, on=True):
        if self.vcomm is not None:
            self.engine._l.vpic_hip_comm_timing(self.vcomm, 1 if on else 0, None, None, None)
        self.trace = on
        self._tr_pairs, self._tr_waits = [], []
        self._tr_host = dict(step=0.0, blocked=0.0, staged=0.0, steps=0)

    def trace_report(self):
        """Per step and rank (call after the device is idle): host time spent issuing the step (its one blocking
        read-back taken out), time the transfers took on the communication stream, time the engine's stream waited for
        them, and the fraction of the transfer time that was hidden behind kernels."""
        n = max(self._tr_host["steps"], 1)
        xfer = sum(a.elapsed_time(b) for a, b in self._tr_pairs) if self._tr_pairs else 0.0
        wait = sum(a.elapsed_time(b) for a, b in self._tr_waits) if self._tr_waits else 0.0
        n_msg = len(self._tr_pairs)
        if self.vcomm is not None:
            import ctypes as C
            x, w, k = C.c_double(), C.c_double(), C.c_int64()
            self.engine._l.vpic_hip_comm_timing(self.vcomm, -1, C.byref(x), C.byref(w), C.byref(k))
            xfer, wait, n_msg = x.value, w.value, k.value
        staged = self._tr_host["staged"] * 1e3
        if self.vcomm is None and (self.comm is None or self.staged):   # a blocking transport: nothing is hidden
            xfer, wait = xfer + staged, wait + staged
        return dict(host_issue_ms_per_step=(self._tr_host["step"] - self._tr_host["blocked"] - self._tr_host["staged"]) * 1e3 / n,
                    host_blocked_ms_per_step=self._tr_host["blocked"] * 1e3 / n,
                    exchange_ms_per_step=xfer / n, exchange_exposed_ms_per_step=wait / n,
                    overlap_frac=(1.0 - wait / xfer) if xfer > 0 else None,
                    messages_per_step=(n_msg / n) if n_msg else None,
                    recovery_rounds=self.n_recovery, reserves=self.n_reserved)

    # a message travelling in direction d (0..2: towards -x, -y, -z; 3..5: towards +) goes to this peer / comes from that one
    def _to(self, d):
        return self.face_rank[d]

    def _from(self, d):
        return self.face_rank[(d + 3) % 6]

    def _exchange(self, send, recv):
        """send/recv: {direction: tensor}; a direction absent from a dict has nothing to move (both
        ends know: the counts went first).  Posts all directions at once; where an axis has two domains both
        peers are the same rank and the fixed order (low face first) keeps sends and receives matched."""
        if not send and not recv:
            return
        if self.vcomm is not None:                           # stream-ordered both ways: nothing for the host to wait for
            self._finish(self._vstart(send, recv))
            return
        self.engine.sync()                                   # packs ran on the engine's stream
        if self.resident:
            self.n_sync_transport += 1                       # a blocking transport's own (gloo rehearsals); none with RCCL
        else:
            self.n_sync += 1
        staged = self.staged
        if staged:
            # gloo moves host memory only: stage through the host (rehearsals on a one-GPU box)
            dev_recv = recv
            send = {d: t.cpu() for d, t in send.items()}
            recv = {d: torch.empty_like(t, device="cpu") for d, t in dev_recv.items()}
        ops = []
        for d in range(6):
            if d in send:
                ops.append(dist.P2POp(dist.isend, send[d], self._to(d), group=self.group))
        for d in range(6):
            if d in recv:
                ops.append(dist.P2POp(dist.irecv, recv[d], self._from(d), group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if staged:
            for d, t in recv.items():
                dev_recv[d].copy_(t)
        if self.dev.type == "cuda":
            torch.cuda.current_stream().synchronize()        # unpacks run on the engine's stream

    def _ensure_inj(self, n):
        if n <= self.inj_cap:
            return
        self.inj_cap = int(n * 1.3) + 1024
        # 48-byte injectors as 12 x int32 (src/species_advance/species_advance.h:48-55)
        self.inj = {(kind, d): torch.empty((self.inj_cap, 12), dtype=torch.int32, device=self.dev)
                    for kind in ("send", "recv") for d in self.dirs}

    def _axis_dirs(self, a):
        return (a, a + 3)

    def exchange_tang_b(self):
        """remote.c:61-134: all shared faces at once (ghost planes need no edge propagation)."""
        e = self.engine
        for d in self.dirs:
            e.pack_tang_b(d, self.fbuf[("send", d)].data_ptr())
        self._exchange({d: self.fbuf[("send", d)] for d in self.dirs}, {d: self.fbuf[("recv", d)] for d in self.dirs})
        for d in self.dirs:
            e.unpack_tang_b(d, self.fbuf[("recv", d)].data_ptr())

    def synchronize_jf(self):
        e = self.engine
        e.local_adjust_jf()
        # x, then y, then z: edges and corners propagate (remote.c:284-289); within an axis both planes are packed before
        # either is accumulated into (remote.c:477-484)
        for a in range(3):
            if a in self.axes:
                dd = self._axis_dirs(a)
                for d in dd:
                    e.pack_jf(d, self.fbuf[("send", d)].data_ptr())
                self._exchange({d: self.fbuf[("send", d)] for d in dd}, {d: self.fbuf[("recv", d)] for d in dd})
                for d in dd:
                    e.unpack_jf(d, self.fbuf[("recv", d)].data_ptr())
            else:
                e.synchronize_jf_self(a)

    def boundary_p(self):
        e = self.engine
        for _ in range(NUM_COMM_ROUND):
            ns = e.boundary_p_pack()
            for a in self.axes:                             # the reference posts all six faces at once; the axes are independent
                dd = self._axis_dirs(a)
                # counts first, payload second (boundary_p.c:341-384)
                for d in dd:
                    self.cnt_send[d][0] = ns[d]
                self._exchange({d: self.cnt_send[d] for d in dd}, {d: self.cnt_recv[d] for d in dd})
                nr = {d: int(self.cnt_recv[d].item()) for d in dd}
                self.n_sync += 1
                self._ensure_inj(max(max(ns[d] for d in dd), max(nr.values())))
                for d in dd:
                    if ns[d]:
                        e.get_injectors(d, self.inj[("send", d)].data_ptr())
                self._exchange({d: self.inj[("send", d)][:ns[d]] for d in dd if ns[d]},
                               {d: self.inj[("recv", d)][:nr[d]] for d in dd if nr[d]})
                for d in dd:
                    if nr[d]:
                        e.boundary_p_inject(self.inj[("recv", d)].data_ptr(), nr[d])
            # the reference always makes num_comm_round rounds (advance.cxx:94-96); a round in which no
            # domain has a mover left does nothing, so stop as soon as that is known (one tiny all-reduce
            # instead of a pack and two exchanges per spared round)
            if _ + 1 < NUM_COMM_ROUND and self._allsum([sum(e.nm(sp) for sp in self.species)])[0] == 0:
                break

    # ---- divergence cleaning family across slabs (advance.cxx:151-208, initialize.cxx:32-76) --------
    def _plane_exchange(self, a, n, pack, unpack):
        """One face message of n floats each way along axis a: pack, exchange, unpack; returns what unpack returns, summed."""
        key = ("plane", a, n)
        dd = self._axis_dirs(a)
        if key not in self.inj:
            self.inj[key] = {(kind, d): torch.empty(n, dtype=torch.float32, device=self.dev) for kind in ("send", "recv") for d in dd}
        b = self.inj[key]
        for d in dd:
            pack(d, b[("send", d)].data_ptr())
        self._exchange({d: b[("send", d)] for d in dd}, {d: b[("recv", d)] for d in dd})
        return sum(unpack(d, b[("recv", d)].data_ptr()) or 0.0 for d in dd)

    def _allsum(self, vals):
        if self.world == 1:
            return [float(v) for v in vals]
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if (self.staged or self.dev.type == "cpu") else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.n_sync += 1
        return [float(v) for v in t]

    def synchronize_rho(self):                              # remote.c:533-622, x then y then z
        e = self.engine
        e.local_adjust_rho()
        for a in range(3):
            if a in self.axes:
                self._plane_exchange(a, e.rho_count(a), e.pack_rho, e.unpack_rho)
            else:
                e.synchronize_rho_self(a)

    def _message(self, kind, a=None):
        """A face message of `kind` along axis a, or along every shared axis."""
        e = self.engine
        return sum(self._plane_exchange(ax, e.message_count(kind, ax), lambda d, p: e.pack_message(kind, d, p),
                                        lambda d, p: e.unpack_message(kind, d, p))
                   for ax in (self.axes if a is None else [a]))

    def compute_div_e_err(self):                            # compute_div_e_err.c:72-207
        self._message(0)
        self.engine.compute_div_e_err()

    def compute_rhob(self):                                 # compute_rhob.c:74-206
        self._message(0)
        self.engine.compute_rhob()

    def clean_div_b(self):                                  # clean_div_b.c:79-247 (div_b_err ghosts first)
        self._message(1)
        self.engine.clean_div_b()

    def compute_curl_b(self):                               # compute_curl_b.c:78-318
        self.exchange_tang_b()                              # advance_e() fills the same ghosts again later; harmless
        self.engine.compute_curl_b()

    def synchronize_tang_e_norm_b(self):                    # remote.c:298-414, returns the global error
        e = self.engine
        e.local_adjust_tang_e_norm_b()
        err = 0.0
        for a in range(3):
            err += self._message(2, a) if a in self.axes else e.synchronize_tang_e_norm_b_self(a)
        return self._allsum([err])[0]

    def compute_rms_div_e_err(self):                        # compute_rms_div_e_err.c:156-159
        s0, s1 = self._allsum(self.engine.rms_div_e_err_local())
        return float(self.grid.eps0) * (s0 / s1) ** 0.5

    def compute_rms_div_b_err(self):
        s0, s1 = self._allsum(self.engine.rms_div_b_err_local())
        return float(self.grid.eps0) * (s0 / s1) ** 0.5

    def accumulate_rho(self):                               # advance.cxx:155-158
        e = self.engine
        e.clear_rhof()
        for sp in self.species:
            e.accumulate_rho_p(sp)
        self.synchronize_rho()

    def clean_div_e_pass(self):                             # advance.cxx:151-173
        self.accumulate_rho()
        self.compute_div_e_err()
        err = self.compute_rms_div_e_err()
        if err > 0:
            self.engine.clean_div_e()
            self.compute_div_e_err()
            err = self.compute_rms_div_e_err()
            if err > 0:
                self.engine.clean_div_e()
        return err

    def clean_div_b_pass(self):                             # advance.cxx:177-195
        e = self.engine
        e.compute_div_b_err()
        err = self.compute_rms_div_b_err()
        if err > 0:
            self.clean_div_b()
            e.compute_div_b_err()
            err = self.compute_rms_div_b_err()
            if err > 0:
                self.clean_div_b()
        return err

    def initialize_fields(self):
        """initialize.cxx:32-76: checks and derived fields of the initial state (before uncenter_p)."""
        e = self.engine
        self.synchronize_tang_e_norm_b()
        e.compute_div_b_err()
        self.compute_rms_div_b_err()
        self.clean_div_b()
        self.compute_curl_b()
        self.accumulate_rho()
        self.compute_rhob()
        self.compute_div_e_err()
        if self.compute_rms_div_e_err() > 0:
            e.clean_div_e()
        self.synchronize_tang_e_norm_b()

    def _msg(self, kind, d, cap, tag):
        key = (kind, d, tag)
        need = Engine.exchange_message_bytes(cap) // 4
        if key not in self.msg or self.msg[key].numel() < need:
            self.msg[key] = torch.zeros(need + need // 4, dtype=torch.int32, device=self.dev)
            if self.dev.type == "cuda":
                torch.cuda.synchronize()                     # the fill ran on torch's stream, the engine writes on its own
        return self.msg[key][:need]

    def _cap(self, kind, d, k):
        if (kind, d, k) not in self.cap:
            self.cap[(kind, d, k)] = self._round_cap(self.deck.get("exchange_cap0", self.plane[d] // 8))
        return self.cap[(kind, d, k)]

    def _round(self, tag, cs, cr, mover_cap, species=None):
        """Pack the movers of `species` (default: all) into one message per shared face, start the transfer; returns
        what _land needs.  cs / cr: capacities {direction: injectors} of the messages sent / received."""
        e = self.engine
        ms = {d: self._msg("send", d, cs[d], tag) for d in cs}
        mr = {d: self._msg("recv", d, cr[d], tag) for d in cr}
        ptrs, caps = [0] * 6, [0] * 6
        for d in cs:
            ptrs[d], caps[d] = ms[d].data_ptr(), cs[d]
        if self.loopback is not None:
            self.loopback.before_pack(self, ms)
        e.exchange_pack(ptrs, caps, mover_cap, species)
        return self._start(ms, mr), ms, mr, cs, cr

    def _land(self, rnd):
        """The received messages of a round join their species (the engine's stream waits for the transfer first)."""
        tok, ms, mr, cs, cr = rnd
        self._finish(tok)
        for d in mr:
            self.engine.exchange_inject(mr[d].data_ptr(), cr[d])

    def push_and_exchange(self):
        """advance_p of every species and boundary_p, overlapped (north star: "boundary-particle exchange ... overlapped
        with interior push on a second HIP stream"; the reference's begin / interior / end pattern of advance_e.c:114,153,
        191-197 applied to boundary_p.c:341-384).  Per species: push the tiles on the shared faces (and the particles that
        arrived since the sort), pack the species' movers into one fixed-capacity message per shared face, start the
        transfer on the communication stream, push the interior tiles behind it -- so species k is on the wire while its
        own interior and the next species are pushed.  Then the arrivals join their species, and the later rounds (all
        species in one small message per face: stragglers that the interior launches left on a face, particles that an
        earlier round delivered onto yet another boundary; one round more than there are cut axes, three at most like the
        reference's num_comm_round) follow.  Counts stay on the device; ONE read-back (engine.exchange_finish) ends the
        step's exchange."""
        e = self.engine
        e.exchange_begin()
        mover_cap = self.mover_cap or (1 << 30)
        rounds = min(len(self.axes) + 1, NUM_COMM_ROUND)
        phased = hasattr(e, "advance_p_phase") and not self.deck.get("no_overlap", False)
        flights, log = [], []
        for k, sp in enumerate(self.species):
            if phased:
                e.advance_p_phase(sp, 1)
            else:
                e.advance_p_async(sp)
            r = self._round(("s", k), {d: self._cap("send", d, k) for d in self.dirs}, {d: self._cap("recv", d, k) for d in self.dirs},
                            mover_cap, species=[sp])
            if phased:
                e.advance_p_phase(sp, 2)                    # the interior, while the message is on the wire
            flights.append(r)
            log.append((("s", k), r))
        for r in flights:
            self._land(r)
        for rnd in range(1, rounds):
            r = self._round(("r", rnd), {d: self.cap2.get(("send", d), self._round_cap(0)) for d in self.dirs},
                            {d: self.cap2.get(("recv", d), self._round_cap(0)) for d in self.dirs}, mover_cap)
            self._land(r)
            log.append((("r", rnd), r))
        H = self._read_back(log)
        # capacities of the next step, from what each sender wanted to send in this one (both ends read the same header)
        per_species = [0] * len(self.species)
        for d in self.dirs:
            for k in range(len(self.species)):
                for kind in ("send", "recv"):
                    self.cap[(kind, d, k)] = self._next_cap(self.cap[(kind, d, k)], H[(kind, ("s", k), d)][1])
                per_species[k] += H[("send", ("s", k), d)][1]
            for rnd in range(1, rounds):
                for kind in ("send", "recv"):                # (per directed message, like the first round's: both of its ends read the same header)
                    w = H[(kind, ("r", rnd), d)][1]
                    if w > self.cap2.get((kind, d), self._round_cap(0)) // 2:
                        self.cap2[(kind, d)] = max(self.cap2.get((kind, d), self._round_cap(0)), self._round_cap(4 * w))
        self.mover_cap = max(65536, 2 * max(per_species + [0]) + 4096)   # the movers of a species leave through ALL its shared faces
        self._recover(H, log)
        self._make_room()

    def _read_back(self, log):
        """engine.exchange_finish over the messages of `log`: {(kind, tag, direction): [count, wanted, 0, 0]}."""
        order = [(kind, tag, d) for tag, r in log for kind in ("recv", "send") for d in (r[2] if kind == "recv" else r[1])]
        ptr = {("send", tag): r[1] for tag, r in log}
        ptr.update({("recv", tag): r[2] for tag, r in log})
        t0 = time.perf_counter()
        hdr = self.engine.exchange_finish([ptr[(kind, tag)][d].data_ptr() for kind, tag, d in order])
        if self.trace:
            self._tr_host["blocked"] += time.perf_counter() - t0
        self.n_sync += 1
        return dict(zip(order, hdr))

    def _recover(self, H, log):
        """A message that was full left its movers parked on their lists, the particles untouched (the reference grows its
        buffers instead, boundary_p.c:131-150, 416-448; here both ends must know a message's size beforehand).  Both ends
        of such a message read the same header -- wanted > count -- so exactly the two ranks concerned run an extra round
        over that face with a message as large as was wanted."""
        e = self.engine
        for attempt in range(4):
            left = sum(e.nm(sp) for sp in self.species)
            need_s = {d: max([h[1] for (kind, tag, dd), h in H.items() if kind == "send" and dd == d] + [0]) for d in self.dirs}
            need_r = {d: max([h[1] for (kind, tag, dd), h in H.items() if kind == "recv" and dd == d] + [0]) for d in self.dirs}
            over_s = {d: self._round_cap(n + 1) for d, n in need_s.items()
                      if any(h[1] > h[0] for (kind, tag, dd), h in H.items() if kind == "send" and dd == d)}
            over_r = {d: self._round_cap(n + 1) for d, n in need_r.items()
                      if any(h[1] > h[0] for (kind, tag, dd), h in H.items() if kind == "recv" and dd == d)}
            if not over_s and not over_r:
                if left:
                    raise RuntimeError("boundary_p: %d movers left after the step's rounds (a particle crossed more domains than "
                                       "that in one step, or more movers than the exchange kernels were launched for: flags %d)"
                                       % (left, getattr(e, "exchange_flags", 0)))
                return
            self.n_recovery += 1
            e.exchange_begin()
            r = self._round(("x", attempt), over_s, over_r, 1 << 30)
            self._land(r)
            log = [(("x", attempt), r)]
            H = self._read_back(log)
        raise RuntimeError("boundary_p: messages kept overflowing")

    def _make_room(self):
        """Keep the species' arrays from running out between sorts: arrivals are appended, departures leave dead slots until
        the next sort.  From 85 % full: sort now when that frees at least 5 % of the array, otherwise enlarge it by
        1.3125 (the reference's growth factor, boundary_p.c:416-448); the mover list likewise."""
        e = self.engine
        if not hasattr(e, "capacity"):
            return
        for sp in self.species:
            extent, max_np, max_nm = e.capacity(sp)
            if extent > 0.85 * max_np:
                if extent - e.np(sp) > 0.05 * max_np:
                    e.sort_p(sp)
                else:
                    e.reserve(sp, int(max_np * 1.3125) + 4096, max_nm)
                    self.n_reserved += 1
            if self.mover_cap and self.mover_cap > 0.7 * max_nm:
                e.reserve(sp, 0, int(max(max_nm, self.mover_cap) * 1.3125) + 4096)
                self.n_reserved += 1

    def step(self, step):
        """vpic_simulation::advance (src/vpic/advance.cxx:38-214) for this domain.  Exchanges are started as soon
        as their payload is packed and finished where their result is needed (advance_e.c:114-197 and
        advance_b.c:111-160 do the same around their begin_/end_ calls); with the RCCL transport what lies between
        runs while the message is on the wire."""
        e, si = self.engine, self.deck.get("sort_interval", 0)
        t_step = time.perf_counter()
        self.n_step += 1
        e.clear_accumulators()
        for sp in self.species:                             # si < 0: adaptive (engine.sort_due), at the latest every -si steps
            if (si > 0 and step % si == 0) or (si < 0 and e.sort_due(sp, -si)):
                e.sort_p(sp)
        if self.resident:
            self.push_and_exchange()
        else:
            for sp in self.species:
                e.advance_p(sp)
            e.reduce_accumulators()
            self.boundary_p()
        if hasattr(e, "clear_jf_unload_accumulator"):
            e.clear_jf_unload_accumulator()                 # advance.cxx:109-110 in one pass
        else:
            e.clear_jf()
            e.unload_accumulator()
        # synchronize_jf: x, then y, then z (edges and corners propagate, remote.c:284-289); within an axis both planes are
        # packed before either is accumulated into (remote.c:477-484).  The first exchange is overlapped with the first
        # half B advance (jf and B are independent).
        e.local_adjust_jf()
        advanced_b = False
        for a in range(3):
            if a not in self.axes:
                e.synchronize_jf_self(a)
                continue
            dd = self._axis_dirs(a)
            for d in dd:
                e.pack_jf(d, self.fbuf[("send", d)].data_ptr())
            tok_jf = self._start({d: self.fbuf[("send", d)] for d in dd}, {d: self.fbuf[("recv", d)] for d in dd})
            if not advanced_b:
                e.advance_b(0.5)
                advanced_b = True
            self._finish(tok_jf)
            for d in dd:
                e.unpack_jf(d, self.fbuf[("recv", d)].data_ptr())
        if not advanced_b:
            e.advance_b(0.5)
        fb = self.fbuf2 if self.resident else self.fbuf     # (a second buffer set while a transport that does not block is at work)
        # tangential-B ghosts of the neighbours (all shared faces at once, remote.c:61-134).  With x the only cut axis the
        # exchange is overlapped with advance_e on the planes x = 2..nx, which need none of the ghosts (advance_e.c:155-327
        # makes the same split); otherwise every plane of the box needs some neighbour's ghosts.
        for d in self.dirs:
            e.pack_tang_b(d, fb[("send", d)].data_ptr())
        tok_b = self._start({d: fb[("send", d)] for d in self.dirs}, {d: fb[("recv", d)] for d in self.dirs})
        split = hasattr(e, "advance_e_part") and self.axes == [0]
        if split:
            e.advance_e_part(1)
        self._finish(tok_b)
        for d in self.dirs:
            e.unpack_tang_b(d, fb[("recv", d)].data_ptr())
        if split:
            e.advance_e_part(2)
        else:
            e.advance_e()
        e.advance_b(0.5)
        ci = self.deck.get("clean_div_e_interval", 0)
        if ci > 0 and step % ci == 0:
            self.clean_div_e_pass()
        ci = self.deck.get("clean_div_b_interval", 0)
        if ci > 0 and step % ci == 0:
            self.clean_div_b_pass()
        ci = self.deck.get("sync_shared_interval", 0)
        if ci > 0 and step % ci == 0:
            self.synchronize_tang_e_norm_b()
        e.load_interpolator()
        if self.trace:
            self._tr_host["step"] += time.perf_counter() - t_step
            self._tr_host["steps"] += 1
