"""Host-side bookkeeping of the asynchronous forward (include/mgsplat.h: mgs_rasterize_forward with async_forward = 1).

The reference blocks every forward on a cudaMemcpy of `num_rendered` (RAST/cuda_rasterizer/rasterizer_impl.cu:284) because
its binning buffer is sized from that count.  Three forward modes (set_forward_mode / MGS_FORWARD_MODE):

  "safe" (default)  a forward can NEVER hand back incomplete images.  A shape whose worst-case workspace fits the budget
                    (set_safe_workspace, default 1 GB: ManiGaussian's own 16 384-Gaussian shape needs 206 MB) is enqueued
                    without any host-device synchronisation -- it cannot overflow; every other shape enqueues everything
                    with a workspace sized from the shape's marks and waits for the PREPROCESS's report only (two pinned
                    words; round 4 waited for the whole render): binning and render are running while the call returns, and
                    a scene that outgrew the buffer is binned and rendered again with room before the call returns.
  "async" (opt-in)  every shape is enqueued without synchronisation: the workspace is sized from what earlier forwards of the
                    same shape needed (high-water marks + head-room) and the device reports {instances, chunk records used,
                    overflow} through two pinned host words.  What HIP-graph capture and a GPU-bound 0.15 ms step need;
                    the price is the overflow protocol below (a scene that outgrows its marks is repaired at backward entry
                    with a RuntimeWarning, or raises late).
  "blocking"        every forward waits for the count (the reference's behaviour, also for small shapes).

This module owns

  * the ring of pinned status slots (one per in-flight forward, tagged so that a stale write is recognisable),
  * the high-water marks per (device, problem shape),
  * the list of forwards whose report has not been read yet.

Reports are read lazily and never waited for on the hot path: at the next forward, at a backward, or on
`check_status()`.  If a scene outgrew its workspace the images (and gradients) of THAT call were incomplete; the first
later call into this module raises RuntimeError (loud, late) after raising the marks so that a re-run fits.  The first
forward of a shape, `debug=True` settings and `set_forward_mode("blocking")` take the blocking path, which cannot overflow.
"""
import collections
import collections.abc
import ctypes
import os
import threading
import warnings
import weakref

import torch

from . import _lib

NSLOTS = 1024  # status slots per device (four 64-bit words each, three used); far more than forwards can be in flight
SLOT_WORDS = 4
SLOT_BYTES = 8 * SLOT_WORDS

# ---- the compiled binding (csrc/mgs_torch.cpp -> _mgs_torch.so): owns the marks and the hot autograd path when present --------
_EXT = [False]  # False: not looked for yet; None: absent / disabled; else the module


def ext():
    """manigaussian_amd._mgs_torch, or None (not built, MGS_NO_COMPILED=1): the ctypes shim then does everything."""
    e = _EXT[0]
    if e is False:
        e = None
        if not os.environ.get("MGS_NO_COMPILED"):
            try:
                _lib.lib()  # the binding links libmgsplat.so: fail with the loader's message, not the linker's
                from . import _mgs_torch as e
                if e.ABI_VERSION != _lib.ABI_VERSION:
                    raise ImportError(f"_mgs_torch ABI {e.ABI_VERSION} != {_lib.ABI_VERSION}")
            except ImportError as err:
                warnings.warn(f"manigaussian_amd: the compiled binding is not available ({err}); using the ctypes shim "
                              "(same kernels, ~3x the host time per call).  Build it with `make -C manigaussian_amd/csrc ext`.",
                              RuntimeWarning)
                e = None
        _EXT[0] = e
        if e is not None:
            _push_config()
            _lib.push_options()
            e.set_python_drain(_drain_python_side)
    return e


def _push_config():
    e = _EXT[0]
    if e:
        e.configure({"safe": 0, "async": 1, "blocking": 2}[_MODE], {"repair": 0, "raise": 1}[_OVERFLOW],
                    _HEADROOM["instances"], _HEADROOM["chunks"], -1 if _SAFE_BYTES is None else _SAFE_BYTES)


def _drain_python_side(index):
    """Called by the compiled binding: at its first forward on a device (this side's state -- a pinned allocation -- must
    exist before anybody captures a HIP graph: capture forbids the allocation) and while this side has unread reports."""
    device_state(torch.device("cuda", index)).drain(_from_ext=True)

_WORDS = {}  # slot pointer -> (numpy view of its device's status block, index of the slot's first word)
_MODE = os.environ.get("MGS_FORWARD_MODE", "safe")  # "safe" | "async" | "blocking"
_STATES = {}
_LOCK = threading.Lock()
_TAG = [0]


def set_forward_mode(mode: str):
    """"safe" (default): no host-device synchronisation for shapes whose worst-case workspace fits the budget (they cannot
    overflow), the reference's blocking read of the instance count for every other shape -- a forward never returns
    incomplete images; "async": no synchronisation for any shape, workspaces sized from earlier calls (opt-in: see the
    module docstring for what an overflow then means); "blocking": every forward waits like the reference does.
    Returns the previous mode."""
    global _MODE
    if mode not in ("safe", "async", "blocking"):
        raise ValueError("forward mode is 'safe', 'async' or 'blocking'")
    old, _MODE = _MODE, mode
    _push_config()
    return old


def forward_mode() -> str:
    return _MODE


# "async" mode only: what the backward does when it learns that ITS forward outgrew the workspace.  "repair": render the
# forward again on the blocking path into the same output tensors, warn, go on -- the gradients are those of the complete
# render, but whatever the caller computed from the incomplete images in between (the loss VALUE, its cotangents) is not
# repaired: the step mixes the two.  "raise": RuntimeError at backward entry; the step is lost, nothing inconsistent is
# applied.  (The default mode "safe" never gets here.)
_OVERFLOW = os.environ.get("MGS_OVERFLOW_POLICY", "repair")


def set_overflow_policy(policy: str):
    """"repair" (default) or "raise": see above.  Returns the previous policy."""
    global _OVERFLOW
    if policy not in ("repair", "raise"):
        raise ValueError("overflow policy is 'repair' or 'raise'")
    old, _OVERFLOW = _OVERFLOW, policy
    _push_config()
    return old


def overflow_policy() -> str:
    return _OVERFLOW


def lazy_allowed(cannot_overflow: bool) -> bool:
    """May a forward be enqueued without waiting for its instance count?  "async": always (marks permitting); "safe": only
    with a workspace that cannot overflow; "blocking": never."""
    return _MODE == "async" or (_MODE == "safe" and cannot_overflow)


# Head-room of an asynchronous forward's workspace over the high-water marks of its shape: the scene may grow by these
# factors from one call to the next before a call overflows (and raises, late).  Scenes that differ wildly between calls of
# one shape (a parity sweep, a data loader mixing scenes) want more, or set_forward_mode("blocking").
_HEADROOM = {"instances": float(os.environ.get("MGS_HEADROOM_INSTANCES", 1.5)),
             "chunks": float(os.environ.get("MGS_HEADROOM_CHUNKS", 2.0))}


# A shape whose WORST-CASE workspace (every Gaussian in every tile, every chunk of every block visited) stays below this many
# bytes is always given that workspace: its forwards cannot overflow, need no warm-up call, no marks and NO wait of any kind.
# Default (round 5): 1/32 of the device's memory, at least 1 GB -- 9 GB on a 288 GB MI355X: ManiGaussian's own shape (16 384
# Gaussians, 128 x 128, 3 feature channels) needs 206 MB, BASELINE configs[1] / [2] (100 000 Gaussians, 128 x 128) 4.0 GB per
# forward in flight; the configs[4] shape (500 000 at 256 x 256: 79 GB) waits for the preprocess's report instead.  Rounds 2-4
# used a fixed 1 GB: configs[2] then went by the marks, and under default options its host ran in lock step with the device
# (one wait per forward; up to +12 % on a slow host, profiles/r05_bench_c2.json).  MGS_SAFE_WORKSPACE_MB / set_safe_workspace
# fix the budget in megabytes (0: never allocate the worst case).
_SAFE_BYTES = (int(float(os.environ["MGS_SAFE_WORKSPACE_MB"]) * (1 << 20)) if os.environ.get("MGS_SAFE_WORKSPACE_MB") else None)
_AUTO_SAFE = {}  # device index -> max(1 GB, total memory / 32)
SAFE_FRACTION = 32


def set_safe_bytes(nbytes):
    """The budget in bytes; None: the default (1/32 of the device's memory, at least 1 GB).  Returns the previous setting."""
    global _SAFE_BYTES
    old, _SAFE_BYTES = _SAFE_BYTES, (None if nbytes is None else int(nbytes))
    _push_config()
    return old


def set_safe_workspace(megabytes):
    """Largest worst-case workspace (MB) a forward simply allocates instead of sizing it from earlier calls / the device's
    report (None: the default, 1/32 of the device's memory)."""
    set_safe_bytes(None if megabytes is None else int(float(megabytes) * (1 << 20)))


def safe_bytes(dev=None) -> int:
    if _SAFE_BYTES is not None:
        return _SAFE_BYTES
    idx = -1
    if torch.cuda.is_available():
        idx = dev.index if (dev is not None and getattr(dev, "index", None) is not None) else torch.cuda.current_device()
    v = _AUTO_SAFE.get(idx)
    if v is None:
        total = torch.cuda.get_device_properties(idx).total_memory if idx >= 0 else 0
        v = _AUTO_SAFE[idx] = max(1 << 30, total // SAFE_FRACTION)
    return v


# ---- worst-case workspaces that live forwards still hold: the budget is charged against their SUM per device -----------------
# (round 5 tested it per call: V forwards before the first backward pinned V x 4 GB at BASELINE configs[2].)  The compiled
# binding keeps the counter when it is loaded (its own forwards charge it too); else this dict does.
_HELD = {}


def held_bytes(index: int) -> int:
    """Bytes of worst-case workspaces live forwards hold on device `index`."""
    e = _EXT[0]
    return int(e.held_bytes(index)) if e else _HELD.get(index, 0)


def _hold_add(index, delta):
    e = _EXT[0]
    if e:
        e.hold_add(index, int(delta))
    else:
        with _LOCK:
            _HELD[index] = _HELD.get(index, 0) + int(delta)


def hold(index: int, nbytes: int, owner):
    """Charge `nbytes` against device `index`'s budget until `owner` (the workspace tensor) is freed."""
    _hold_add(index, nbytes)
    weakref.finalize(owner, _hold_add, index, -int(nbytes))


def worst_case_fits(index: int, nbytes: int, dev=None) -> bool:
    """May a forward take a worst-case workspace of `nbytes` now?  (what is held + this one) <= the budget."""
    return held_bytes(index) + int(nbytes) <= safe_bytes(dev)


def set_headroom(instances: float = None, chunks: float = None):
    """Factors (>= 1) by which an asynchronous forward's instance list / chunk-record pool exceed the largest count seen."""
    for k, v in (("instances", instances), ("chunks", chunks)):
        if v is not None:
            if not v >= 1.0:
                raise ValueError("head-room factors are >= 1")
            _HEADROOM[k] = float(v)
    _push_config()


class Pending:
    """One forward whose device report has not been read yet."""
    __slots__ = ("a", "V", "slot_ptr", "key", "num_rendered", "chunks_used", "ref_rendered", "rc", "captured", "recoverable",
                 "recovered", "backward_enqueued", "tag", "cap", "pool", "words")

    def __init__(self, a, V, slot_ptr, key, captured=False, recoverable=False):
        self.a, self.V, self.slot_ptr, self.key = a, V, slot_ptr, key
        self.words = _WORDS.get(slot_ptr)  # (numpy view of the status block, index of word 0) for a cheap "anything yet?"
        self.num_rendered = self.chunks_used = -1   # instances BINNED (what sizes a workspace), chunk records used
        self.ref_rendered = -1                      # the reference's num_rendered (3-sigma rects), status word 2
        self.rc = _lib.MGS_PENDING
        self.captured = captured
        self.recoverable = recoverable    # a backward will follow and can re-render (manigaussian_amd._C.recover_forward)
        self.recovered = False
        self.backward_enqueued = False    # ... but it has been enqueued on the incomplete state already
        # `a` is shared with the forward's handle and is rewritten by a recovery: remember what THIS run was given
        self.tag, self.cap, self.pool = int(a.status_tag), int(a.binning_capacity), int(a.chunk_pool)

    def poll(self):
        """Non-blocking read of the status words; returns the library's code (MGS_PENDING until both words arrived)."""
        if self.rc != _lib.MGS_PENDING:
            return self.rc
        if self.recovered:  # its arguments describe the re-run now; this run's verdict is in
            return self.rc
        w = self.words
        if w is not None and w[0][w[1]] == -1 and w[0][w[1] + 1] == -1:  # both words still "pending": no call into the library
            return self.rc
        L = _lib.lib()
        nr, ch, ref = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(-1)
        if self.V:
            rc = L.mgs_forward_result_views(ctypes.byref(self.a), self.V, self.slot_ptr, ctypes.byref(nr), ctypes.byref(ch),
                                            ctypes.byref(ref))
        else:
            rc = L.mgs_forward_result(ctypes.byref(self.a), self.slot_ptr, ctypes.byref(nr), ctypes.byref(ch), ctypes.byref(ref))
        if nr.value >= 0:
            self.num_rendered = nr.value
        if ch.value >= 0:
            self.chunks_used = ch.value
        if ref.value >= 0:
            self.ref_rendered = ref.value
        if rc != _lib.MGS_PENDING:
            self.rc = rc
        return rc


class _Marks(collections.abc.MutableMapping):
    """The marks of one device when the compiled binding is loaded: ONE store, in the binding, shared by its hot path and by
    this shim (a shape warmed up through either is known to both).  Values are [instances, chunk records or None] lists,
    copies -- assign to change one."""

    def __init__(self, e, index):
        self.e, self.index = e, index

    def __getitem__(self, key):
        v = self.e.marks_get(self.index, tuple(key))
        if v is None:
            raise KeyError(key)
        return v

    def __setitem__(self, key, value):
        self.e.marks_set(self.index, tuple(key), int(value[0]), None if value[1] is None else int(value[1]))

    def __delitem__(self, key):
        if not self.e.marks_del(self.index, tuple(key)):
            raise KeyError(key)

    def __iter__(self):
        return iter(self.e.marks_keys(self.index))

    def __len__(self):
        return len(self.e.marks_keys(self.index))


class DeviceState:
    def __init__(self, dev):
        self.dev = dev
        self.index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.status = torch.full((SLOT_WORDS * NSLOTS,), -1, dtype=torch.int64).pin_memory()
        self.base_ptr = self.status.data_ptr()
        words = self.status.numpy()
        for i in range(NSLOTS):
            _WORDS[self.base_ptr + SLOT_BYTES * i] = (words, SLOT_WORDS * i)
        self.reserved = set()  # slots held by captured graphs
        self.next_slot = 0
        # shape key -> [instances high-water, chunk records high-water or None (unknown: worst case)]
        e = ext()
        self.marks = _Marks(e, self.index) if e else {}
        self.pending = collections.deque()
        self.captured = []   # forwards recorded into HIP graphs: their slots stay reserved, check_status() reads them
        self.deferred = collections.deque(maxlen=64)  # overflowed forwards whose backward will re-render (diagnostics)
        self.lock = threading.Lock()

    def take_slot(self):
        """(pointer to two pinned words, tag) for one forward."""
        with self.lock:
            reserved = self.reserved
            for _ in range(NSLOTS):
                i = self.next_slot
                self.next_slot = (i + 1) % NSLOTS
                ptr = self.base_ptr + SLOT_BYTES * i
                if not reserved or ptr not in reserved:
                    break
            else:
                raise RuntimeError("all status slots are held by captured graphs")
            _TAG[0] = tag = (_TAG[0] + 1) & 0xffff  # (unique per in-flight forward of this device is all that is needed)
        return ptr, tag

    def add(self, pending):
        """Register a forward whose report is outstanding (thread-safe: ctypes calls release the GIL)."""
        with self.lock:
            if pending.captured:
                self.captured.append(pending)
                self.reserved.add(pending.slot_ptr)
            else:
                self.pending.append(pending)
                e = _EXT[0]
                if e:  # the binding's hot path looks at this side's reports too while some are outstanding
                    e.set_python_pending(self.index, len(self.pending))

    # ---- high-water marks ---------------------------------------------------------------------------------------
    def guess(self, key):
        """(capacity, chunk pool) for an asynchronous forward of this shape, or None if the shape is new."""
        m = self.marks.get(key)
        if m is None or m[1] is None:
            return None
        return int(m[0] * _HEADROOM["instances"]) + 4096, int(m[1] * _HEADROOM["chunks"]) + 64

    def learn(self, key, R=None, chunks=None, pool_unknown=False):
        m = list(self.marks.get(key) or (0, None))
        if R is not None and R > m[0]:
            m[0] = R
        if pool_unknown:
            m[1] = None
        elif chunks is not None and (m[1] is None or chunks > m[1]):
            m[1] = chunks
        self.marks[key] = m

    # ---- reports --------------------------------------------------------------------------------------------------
    def drain(self, wait=False, _from_ext=False):
        """Read every report that has arrived (wait=True: all of them, synchronising with the device once if one is still
        outstanding).  Raises if a forward overflowed its workspace or finished without reporting."""
        e = _EXT[0]
        if e and not _from_ext:  # reports of forwards the compiled binding enqueued on this device
            e.check_status(self.index, wait)
        if not self.pending:
            return
        failed = None
        synced = False
        with self.lock:  # the deque is only ever mutated under the lock and in place (add() appends under it too)
            todo = list(self.pending)
            self.pending.clear()
            keep = []
            for i, p in enumerate(todo):
                rc = p.poll()
                if rc == _lib.MGS_PENDING and (wait or (len(todo) - i - 1) + len(keep) >= NSLOTS // 2):
                    if not synced:  # everything enqueued so far has run after this: a report that is still missing never comes
                        torch.cuda.synchronize(self.dev)
                        synced = True
                    rc = p.poll()
                    if rc == _lib.MGS_PENDING:
                        failed = failed or "a rasterizer forward finished without reporting its instance count"
                        continue
                if rc == _lib.MGS_PENDING:
                    keep.append(p)
                    continue
                failed = self._account(p, rc) or failed
            self.pending.extendleft(reversed(keep))  # ahead of anything appended meanwhile (nothing: we hold the lock)
            if e:
                e.set_python_pending(self.index, len(self.pending))
        if failed:
            raise RuntimeError(failed)

    def _account(self, p, rc):
        """Fold a finished forward into the marks; returns an error message if it overflowed."""
        if rc == _lib.MGS_OK:
            self.learn(p.key, p.num_rendered, p.chunks_used)
            return None
        if rc in (_lib.MGS_NEED_CAPACITY, _lib.MGS_RETRY_TABLE_INIT):
            grew = rc == _lib.MGS_NEED_CAPACITY
            if grew:
                over_inst = p.num_rendered > p.cap > 0
                self.learn(p.key, p.num_rendered if p.num_rendered >= 0 else None, None, pool_unknown=not over_inst)
                what = (f"{p.num_rendered} (Gaussian, tile) instances > capacity {p.cap}" if over_inst else
                        f"chunk pool of {p.pool} records")
                msg = ("an asynchronous rasterizer forward outgrew the workspace sized from earlier calls of the same shape "
                       f"({what}): the images of THAT call were incomplete.  The marks are raised")
                hint = ("  For scenes that grow abruptly use manigaussian_amd.set_forward_mode('safe') (the default) or a larger "
                        "set_headroom().")
                hint2 = ", or use manigaussian_amd.set_forward_mode('safe') (the default) for scenes that grow abruptly."
            else:  # nothing to learn: a workgroup of the preprocess launch was held back (include/mgsplat.h MGS_RETRY_TABLE_INIT)
                msg = ("an asynchronous rasterizer forward's preprocess gave up waiting for its zeroed tile tables (a workgroup of "
                       "the launch made no progress for about a second) and binned nothing: the images of THAT call were incomplete")
                hint = "  manigaussian_amd.set_options(table_init=1) removes the hand-shake."
                hint2 = " (manigaussian_amd.set_options(table_init=1) removes the hand-shake)."
            if p.recovered or (_OVERFLOW == "repair" and p.recoverable and not p.backward_enqueued and not p.captured):
                # its backward re-renders first (manigaussian_amd._C.recover_forward): the gradients come from a complete
                # forward; only what the caller computed from the incomplete images in between cannot be repaired
                if not p.recovered:
                    self.deferred.append(p)
                warnings.warn(msg + "; the call's backward re-renders on the blocking path before it runs, but a loss computed "
                              "from those images was computed from incomplete images." + hint, RuntimeWarning, stacklevel=4)
                return None
            if p.recoverable and not p.backward_enqueued and not p.captured:  # (overflow policy "raise")
                p.recovered = True  # its backward, if it still comes, raises too (manigaussian_amd._C._settle)
                return msg + "; the step is lost (overflow policy 'raise'): re-run it" + (hint2 if grew else ".")
            return msg + " and its gradients were computed on the incomplete state; re-run the step" + hint2
        return f"rasterizer forward failed: {_lib.last_error()} (code {rc})"

    def check_captured(self):
        bad = None
        for p in self.captured:
            p.rc = _lib.MGS_PENDING
            rc = p.poll()
            if rc not in (_lib.MGS_OK, _lib.MGS_PENDING):
                bad = self._account(p, rc) or bad
        if bad:
            raise RuntimeError(bad)


def device_state(dev) -> DeviceState:
    st = _STATES.get(dev)
    if st is None:
        with _LOCK:
            st = _STATES.get(dev)
            if st is None:
                st = DeviceState(dev)
                _STATES[dev] = st
    return st


def check_status(device=None, wait=True):
    """Read the outstanding forward reports of `device` (default: all devices used so far), waiting for them unless
    wait=False, and raise if one of them -- or a forward replayed from a captured HIP graph -- overflowed its workspace.
    A training loop calls this wherever it synchronises anyway (logging a loss, an optimizer step)."""
    seen = set()
    for dev, st in list(_STATES.items()):
        if device is not None and torch.device(device) != dev:
            continue
        st.drain(wait=wait)  # (drains the compiled binding's reports of the device too)
        st.check_captured()
        seen.add(dev.index)
    e = _EXT[0]
    if e:  # devices only the compiled binding has used
        if device is None:
            e.check_status(-1, wait)
        else:
            d = torch.device(device)
            idx = d.index if d.index is not None else torch.cuda.current_device()
            if idx not in seen:
                e.check_status(idx, wait)
