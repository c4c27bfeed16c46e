"""`sparseconvnet.SCN` -- the reference's native-extension namespace, re-implemented on the
MI355X C ABI (include/aabr_hip.h).

Function names, argument order and ownership rules are those of the reference's pybind11
module (SparseConvNet/sparseconvnet/SCN/pybind.cpp:11-235; dispatch in
SCN/sparseconvnet_cuda.cpp:281-455): the caller passes an EMPTY output tensor which is
resized and filled here; optional tensors arrive as empty tensors; d_weight / d_bias arrive
pre-zeroed.  What differs by design: all grid state (hash tables, site lists, rule tables)
lives in HBM inside `Metadata_3`, nothing is rebuilt or copied per layer, and every kernel
runs on the current PyTorch HIP stream.
"""
import ctypes as C

import torch

import os

import _hip
from _hip import ptr, stream, check

count_macs = os.environ.get("AABR_COUNT_MACS", "1") != "0"  # the reference returns the multiply-add count of every conv; here the count is
                   # a LAZY number: the rule counts stay on the device until somebody reads it


trace = None  # measurement hook: when set to a list, every conv / dW launch appends
              # (kind, n_in, n_out, gather, rows_in, flags, dtype) -- bench.py uses it to find the dominant kernel


class _TotalsRing(object):
    """Device buffer of per-rule-book rule totals (float64).  A rule book writes its total into the next free
    slot with ONE reduction launch when a layer first asks for it; reading values back is one copy of the
    whole buffer.  When the buffer is full it is read back once, kept on the host under its generation
    number, and reused."""
    CAP = 4096

    def __init__(self, device):
        self.buf = torch.zeros(self.CAP, dtype=torch.float64, device=device)
        self.gen, self.n, self.retired, self._snap = 0, 0, {}, None

    def alloc(self):
        if self.n == self.CAP:
            self.retired[self.gen] = self.buf.tolist()
            self.retired.pop(self.gen - 8, None)
            self.gen, self.n, self._snap = self.gen + 1, 0, None
        self.n += 1
        self._snap = None
        return self.gen, self.n - 1

    def value(self, gen, idx):
        if gen == self.gen:
            if self._snap is None:
                self._snap = self.buf[:self.n].tolist()  # the one read-back (host sync) of a fold
            return self._snap[idx]
        return self.retired[gen][idx]


_rings = {}


def _ring(device):
    r = _rings.get(device)
    if r is None:
        r = _rings[device] = _TotalsRing(device)
    return r


class LazyMacs(object):
    """Sum of (rule-total slot, multiplier) terms that behaves like the float the
    reference returns (`sparseconvnet.forward_pass_multiplyAdd_count += ...`,
    submanifoldConvolution.py:85-94) but performs the device->host read only when the value
    is actually looked at -- so counting costs no host synchronisation in the training loop.
    A term references a slot of the per-device totals buffer (never a rule table); a rule book costs
    one small reduction launch however many layers use it, and the pending list is folded into the
    base value (one read-back) once it holds more than 1024 terms."""
    __slots__ = ("terms", "base")

    def __init__(self, terms=(), base=0.0):
        self.terms, self.base = list(terms), float(base)

    def _value(self):
        if self.terms:
            self.base += sum(ring.value(gen, idx) * m for ring, gen, idx, m in self.terms)
            self.terms = []
        return self.base

    def __add__(self, other):
        if isinstance(other, LazyMacs):
            r = LazyMacs(self.terms + other.terms, self.base + other.base)
        else:
            r = LazyMacs(self.terms, self.base + float(other))
        if len(r.terms) > 1024:
            r._value()
        return r

    __radd__ = __add__

    def __iadd__(self, other):  # `counter += macs` in the layer code: amortised O(1), in place
        if isinstance(other, LazyMacs):
            self.terms.extend(other.terms)
            self.base += other.base
        else:
            self.base += float(other)
        if len(self.terms) > 1024:
            self._value()
        return self

    def __float__(self):
        return self._value()

    def __int__(self):
        return int(self._value())

    def __eq__(self, other):
        return self._value() == float(other)

    def __lt__(self, other):
        return self._value() < float(other)

    def __le__(self, other):
        return self._value() <= float(other)

    def __gt__(self, other):
        return self._value() > float(other)

    def __ge__(self, other):
        return self._value() >= float(other)

    def __hash__(self):
        return hash(self._value())

    def __repr__(self):
        return repr(self._value())

    __str__ = __repr__

    def __format__(self, spec):
        return format(self._value(), spec)


def n_rulebook_bits():
    """pybind.cpp:234"""
    return 32


_key_cache = {}


def _key(t):
    """tuple form of a (small, long-lived, CPU) size tensor; memoised per tensor object/version"""
    if not hasattr(t, "tolist"):
        return tuple(int(x) for x in t)
    k = (id(t), t._version)
    v = _key_cache.get(k)
    if v is None or v[0] is not t:
        if len(_key_cache) > 4096:
            _key_cache.clear()
        v = (t, tuple(int(x) for x in t.tolist()))
        _key_cache[k] = v
    return v[1]


# ------------------------------------------------------------------------------------------------
# geometry plan (extension): inside `with geom_plan():` the rule-book builders below do not launch, they append an
# AabrGeomOp record; the list goes to the library in ONE call (aabr_geom_run) at `flush_geom()` / the end of the block.
# Same entry points, same arguments, same order on the same stream -- only the ~10 us of interpreter + ctypes time per
# launch (x ~250 launches per FPN_Net pass) leave the host's critical path.
# ------------------------------------------------------------------------------------------------
import struct as _struct

_GEOM = _struct.Struct("<ii10i4q8Q")
assert _GEOM.size == 144
G_SUBM, G_TABLES, G_TILE, G_WIDE, G_PAIRS, G_SITES, G_SOFF, G_BK_BUILD, G_BK_SUBM, G_BK_TABLES = \
    1, 2, 3, 4, 5, 7, 8, 9, 10, 11
import threading as _threading


class _GeomTLS(_threading.local):
    rec = None        # None: launch immediately; else [bytearray, count, keep-alive list] -- per THREAD: a prefetch
                      # thread records the next batch's builders while the training thread runs its own


_gtls = _GeomTLS()
geom_stats = {"plans": 0, "ops": 0}


def _p(t):
    return t.data_ptr() if t is not None else 0


def _geom(kind, i32=(), i64=(), ps=()):
    """record (inside a geom_plan block) or launch (outside) one builder call"""
    i32 = list(i32) + [0] * (10 - len(i32))
    i64 = list(i64) + [0] * (4 - len(i64))
    ps = list(ps) + [0] * (8 - len(ps))
    r = _gtls.rec
    if r is not None:
        r[0] += _GEOM.pack(kind, 0, *i32, *i64, *ps)
        r[1] += 1
        return
    buf = C.create_string_buffer(_GEOM.pack(kind, 0, *i32, *i64, *ps), 144)
    check(_hip.load().aabr_geom_run(buf, 1, stream()))


def flush_geom():
    """hand the records collected so far to the library (call before anything reads what they produce)"""
    r = _gtls.rec
    if r is None or r[1] == 0:
        return
    buf = (C.c_char * len(r[0])).from_buffer(r[0])
    n = r[1]
    geom_stats["plans"] += 1
    geom_stats["ops"] += n
    try:
        check(_hip.load().aabr_geom_run(buf, n, stream()))
    finally:
        del buf
        r[0] = bytearray()
        r[1] = 0
        del r[2][:]


class geom_plan(object):
    def __enter__(self):
        self.outer = _gtls.rec
        if self.outer is None:
            _gtls.rec = [bytearray(), 0, []]
        return self

    def __exit__(self, *exc):
        if self.outer is None:
            try:
                if exc[0] is None:
                    flush_geom()
            finally:
                _gtls.rec = None
        return False


def _keep(*ts):
    """scratch tensors of recorded launches stay alive until the plan has been handed over"""
    if _gtls.rec is not None:
        _gtls.rec[2].extend(ts)


_parked = []          # (event on the consumer stream, tensors) of dead Metadata objects whose geometry was built elsewhere
_park_lock = _threading.Lock()


def _reap_handed_over():
    """drop the parked geometry whose consumer stream has passed its last use"""
    with _park_lock:
        while _parked and _parked[0][0].query():
            _parked.pop(0)


class _Brick(object):
    """brick form of one scale (csrc/brick.hip, csrc/geom.h BrickLevel): ONE allocation = dense directory (one 16-byte
    entry per super-brick of the extent) followed by the occupied bricks; `ext` = the extent (largest coordinate + 1 per
    axis) the directory covers, `dims` = (super-bricks per axis x3, samples)."""
    __slots__ = ("level", "dims", "packed", "nw", "nb_cap", "v_cap", "bcoord", "meta", "ext", "coords_full")

    def __init__(self, ext, nsamples, nb_bound, v_bound, device, meta, alloc=True):
        self.ext = tuple(int(e) for e in ext)
        self.dims = tuple(max((e + 15) // 16, 1) for e in self.ext) + (max(int(nsamples), 1),)
        assert all(d <= 4096 for d in self.dims[:3]) and self.dims[3] <= 65535
        self.packed = self.dims[0] | (self.dims[1] << 16) | (self.dims[2] << 32) | (self.dims[3] << 48)
        self.nw = self.dims[0] * self.dims[1] * self.dims[2] * self.dims[3]
        self.v_cap = max(min(int(v_bound), self.nw * 4096), 1)
        self.nb_cap = max(min(int(nb_bound), self.v_cap, self.nw * 64), 1)
        self.level = torch.empty(4 * (self.nw + self.nb_cap), dtype=torch.int32, device=device) if alloc else None
        self.bcoord = torch.empty((self.nb_cap, 4), dtype=torch.int32, device=device)
        self.coords_full = torch.empty((self.v_cap, 4), dtype=torch.int32, device=device)
        self.meta = meta

    def dims_c(self):
        return _hip.i32xn(self.dims)

    def dir_ptr(self):
        return self.level.data_ptr()

    def bricks_ptr(self):
        return self.level.data_ptr() + 16 * self.nw

    def tensors(self):
        return [self.level, self.bcoord, self.coords_full, self.meta]


def _brick_build(src_coords, vin_bound, vin_count_ptr, size, stride, out_sp, bk, flags=0, scratch=None):
    """record / launch aabr_brick_build of level `bk` from the sites `src_coords` (flags: 1 = buffers already cleared by
    the caller)"""
    lib = _hip.load()
    if scratch is None:
        scratch = torch.empty(int(lib.aabr_brick_scratch_words(bk.nw, bk.nb_cap)) + 2, dtype=torch.int32,
                              device=bk.level.device)
    sp = scratch.data_ptr() + 4 * ((-scratch.data_ptr() // 4) % 2)              # 8-byte aligned status words
    _geom(G_BK_BUILD, tuple(size) + tuple(stride) + tuple(out_sp) + (flags,), (vin_bound, bk.packed, bk.nb_cap, bk.v_cap),
          (_p(src_coords), vin_count_ptr, bk.level.data_ptr(), bk.bcoord.data_ptr(), bk.coords_full.data_ptr(),
           bk.meta.data_ptr(), sp))
    _keep(scratch)
    return scratch


class _Grid(object):
    """one scale of the scene: site list + hash table (replaces SparseGrids, Metadata.h:24-34); with
    Metadata_3(site_order="brick") the hash table is replaced by a brick level (`brick`, keys None)"""
    __slots__ = ("coords", "keys", "vals", "cap", "V", "batch_size", "sample_off", "brick")

    def __init__(self, coords, keys, vals, cap, V, sample_off=None, brick=None):
        self.coords, self.keys, self.vals, self.cap, self.V = coords, keys, vals, cap, V
        self.brick = brick
        # host list: first row of sample b (SparseGrid::ctr, Metadata.h:24-33) for b = 0 .. MAX_SAMPLES, or None
        # when the grid was built without the ride-along read
        self.sample_off = sample_off

    def sample_counts(self, nb):
        """sites per sample for samples 0..nb-1 from the offsets read with V, or None"""
        o = self.sample_off
        if o is None or nb + 1 > len(o):
            return None
        return [o[b + 1] - o[b] for b in range(nb)]


# voxel-scatter form (csrc/voxel_scatter.hip): 2 LDS-binned (default), 1 one atomic per point, 0 generic; AABR_SCATTER
# overrides (the A/B of tools/tools_scatter_kernels.py); below SCATTER_MIN_POINTS the generic form's fewer launches win
import os as _os
scatter_variant = int(_os.environ.get("AABR_SCATTER", "2"))
SCATTER_MIN_POINTS = int(_os.environ.get("AABR_SCATTER_MIN_POINTS", "32768"))
scatter_stats = {"variant0": 0, "variant1": 0, "variant2": 0, "redone": 0, "brick": 0}
# brick-major rows: the voxel scatter straight into the brick grid (csrc/brick.hip aabr_points_prepare / aabr_points_sites:
# no hash table, no first-seen numbering); False (AABR_BRICK_SCATTER=0): the hash scatter above followed by a renumbering
# (Metadata_3._brickify_input, the first round-5 form) -- kept for the A/B and as a second implementation to test against
brick_scatter = _os.environ.get("AABR_BRICK_SCATTER", "1") != "0"

def derived_cap(E):
    """slots of a derived (strided-level) grid that receives at most E keys: 1.5 E rounded up to a power of two --
    worst-case load 2/3, typical 0.15-0.35 (a level keeps a half to a sixth of the sites it is derived from).  Round 3
    used 2 E: every derived grid of a pass was as large as its base grid and was cleared every step."""
    return _hip.next_pow2(max(64, E + (E + 1) // 2))


MAX_SAMPLES = 61     # per-sample offsets that ride along with a grid's site-count read (64-word rows)
BRICK_MAX_DIR_WORDS = 1 << 25   # 512 MiB of directory for the input level; above it a scene keeps its hash grids
brick_stats = {"built": 0, "declined": 0}


def _brick_dir_words(extent):
    w = max(extent[3] + 1, 1)
    for e in extent[:3]:
        w *= max((e + 16) // 16, 1)
    return w


_geom_mailboxes = []     # recycled _hip.Mailbox objects of deferred pyramid reads
_BK_MAX_LEVELS = 48  # brick levels of one Metadata (input + strided levels); one read-back row each:
_BK_ROW = 64 + 16    # [V, per-sample offsets .. (64 words) | the level's meta block (16 words)]


class _Gather(object):
    """one gather table [vol, rows] + its compiled streaming forms (built lazily, cached)"""
    __slots__ = ("table", "counts", "vol", "rows", "_blocks", "_blocks256", "_pairs", "_host_counts", "_total")

    def __init__(self, table, counts, vol, rows):
        self.table, self.counts, self.vol, self.rows = table, counts, vol, rows
        self._blocks = self._blocks256 = self._pairs = self._host_counts = self._total = None

    def total_slot(self):
        """(ring, generation, index) of this rule book's rule total on the device"""
        if self._total is None:
            flush_geom()
            ring = _ring(self.table.device)
            gen, idx = ring.alloc()
            torch.sum(self._ensure_counts(), (0,), dtype=torch.float64, out=ring.buf[idx])
            self._total = (ring, gen, idx)
        return self._total

    def _ensure_counts(self):
        if self.counts is None:  # table built without counts (input side of a strided book)
            flush_geom()         # torch reads the table right away
            nb = (self.rows + 255) // 256
            c = (self.table >= 0).to(torch.int32)
            pad = nb * 256 - self.rows
            if pad:
                c = torch.nn.functional.pad(c, (0, pad))
            self.counts = c.view(self.vol, nb, 256).sum(2, dtype=torch.int32).contiguous().view(-1)
        return self.counts

    def blocks(self):
        """tile blocks for aabr_conv_forward"""
        if self._blocks is None:
            lib = _hip.load()
            w = torch.empty(max(lib.aabr_tile_blocks_words(self.rows, self.vol), 1), dtype=torch.int32,
                            device=self.table.device)
            _geom(G_TILE, (self.vol,), (self.rows,), (_p(self.table), _p(w)))
            self._blocks = w
        return self._blocks

    def blocks_wide(self, tile_rows):
        """128- / 256-row tile blocks for aabr_conv_forward_wide (wide layers)"""
        if self._blocks256 is None:
            self._blocks256 = {}
        w = self._blocks256.get(tile_rows)
        if w is None:
            lib = _hip.load()
            w = torch.empty(max(lib.aabr_wide_blocks_words(self.rows, self.vol, tile_rows), 1), dtype=torch.int32,
                            device=self.table.device)
            _geom(G_WIDE, (self.vol, tile_rows), (self.rows,), (_p(self.table), _p(w)))
            self._blocks256[tile_rows] = w
        return w

    def pairs(self):
        """offset-major compacted pairs for aabr_conv_backward_weight"""
        if self._pairs is None:
            lib = _hip.load()
            w = torch.empty(max(lib.aabr_offset_pairs_words(self.rows, self.vol), 1), dtype=torch.int32,
                            device=self.table.device)
            _geom(G_PAIRS, (self.vol,), (self.rows,), (_p(self.table), _p(self._ensure_counts()), _p(w)))
            self._pairs = w
        return self._pairs

    def rule_counts(self):
        """per-offset rule counts (host list); one small D2H read, cached"""
        if self._host_counts is None:
            c = self._ensure_counts()
            flush_geom()
            self._host_counts = c.view(self.vol, -1).sum(1).tolist() if c.numel() else [0] * self.vol
        return self._host_counts

    def max_chunks(self, n_in, n_out):
        """upper bound on the chunks of the weight-gradient pass"""
        cp = _hip.load().aabr_conv_dw_chunk_pairs(self.rows, self.vol, n_in, n_out)
        bound = (self.vol * self.rows + cp - 1) // cp + self.vol
        # the geometric bound sizes a grow-only scratch of bound * n_in * n_out floats (256->256 on a 1.5 M-site
        # book: ~10 GB); past 256 MB read the true per-offset counts once (one small D2H, cached) instead
        if self._host_counts is None and bound * n_in * n_out * 4 > (256 << 20):
            self.rule_counts()
        if self._host_counts is not None:
            return sum((c + cp - 1) // cp for c in self._host_counts)
        return bound


def prefetch_totals(gathers):
    """rule totals of many rule books with ONE launch (LazyMacs slots; `total_slot` otherwise costs one reduction
    launch per rule book at its first use)"""
    todo = [g for g in gathers if g is not None and g._total is None and g.counts is not None and g.rows > 0]
    if not todo:
        return
    flush_geom()
    ring = _ring(todo[0].table.device)
    es = ring.buf.element_size()
    cs, ns, os_ = [], [], []
    for g in todo:
        gen, idx = ring.alloc()
        g._total = (ring, gen, idx)
        cs.append(g.counts.data_ptr())
        ns.append(g.counts.numel())
        os_.append(ring.buf.data_ptr() + idx * es)
    n = len(todo)
    check(_hip.load().aabr_sum_counts((C.c_void_p * n)(*cs), (C.c_int64 * n)(*ns), (C.c_void_p * n)(*os_), n,
                                      stream()))


class _Table(object):
    """rule book: `out` gathers input rows per output row, `inn` gathers output rows per input row
    (for submanifold books `inn` is `out` read with mirrored offsets)"""
    __slots__ = ("out", "inn", "vol", "V_out", "V_in", "flip_ok")

    def __init__(self, out, inn, vol, V_out, V_in, flip_ok=False):
        self.out, self.inn, self.vol = out, inn, vol
        self.V_out, self.V_in, self.flip_ok = V_out, V_in, flip_ok

    def rule_counts(self):
        return self.out.rule_counts()

    def total_rules(self):
        return float(sum(self.rule_counts()))


_pinned_pool = []  # (pinned int32[META_WORDS], event) pairs of finished asynchronous read-backs


class Metadata_3(object):
    """Replaces Metadata<3> (SCN/Metadata/Metadata.h:44-163, pybind class Metadata_3,
    pybind.cpp:12-32,205).  Caches are keyed exactly like the reference's: grids by spatial
    size, submanifold rule books by (spatial, filter), strided rule books by
    (input spatial, filter, stride) (Metadata.cpp:429-443,484-510)."""

    dimension = 3

    def __init__(self, site_order="first_seen"):
        """`site_order` (extension): "first_seen" = the reference's numbering of the input sites (IOLayersRules.h:86-91,
        exact) over hash grids; "brick" = every level numbered brick by brick (csrc/brick.hip): the same sites, rule
        books and features up to a per-sample permutation of the rows, spatial neighbours adjacent in
        memory, no hash tables -- what FPN_Net(site_order="brick") / bench.py run on."""
        assert site_order in ("first_seen", "brick")
        self.site_order = site_order
        self.clear()

    def clear(self):
        self.grids = {}
        self.submanifold = {}
        self.rulebooks = {}
        self.input = None  # dict(point_site, first_pt, cnt_extra, head, nxt, last_pt, meta, n, V, mode, spatial)
        self.device = None
        self._pregrids = set()
        self._brick_rows, self._brick_nrows, self._brick_keep = None, 0, []

    # ---- reference-visible queries ----------------------------------------------------
    def getSpatialLocations(self, spatial_size):
        """LongTensor [V,4] (x,y,z,batch), CPU like the reference (Metadata.cpp:147-168)."""
        return self.getSpatialLocationsDevice(spatial_size).cpu()

    def getSpatialLocationsDevice(self, spatial_size):
        if getattr(self, "_inb", None) is not None:
            self._materialise_input(torch.device("cuda", torch.cuda.current_device()))
        g = self.grids[_key(spatial_size)]
        loc = torch.empty((g.V, 4), dtype=torch.int64, device=g.coords.device)
        check(_hip.load().aabr_spatial_locations(ptr(g.coords), g.V, ptr(loc), stream()))
        return loc

    def hand_over(self, consumer):
        """Geometry built on another stream (FPN_Net.prepare) is about to be used on `consumer`.  Its tensors' memory
        belongs to the producer stream's pool; it must not be reused there before `consumer` has passed every launch
        that reads it.  `tensor.record_stream(consumer)` does that PER TENSOR: at free time the allocator records one
        event on the consumer stream for each of the ~200 geometry tensors of a pass -- 200 marker packets between two
        training steps, measured as 1.07 ms of idle main stream per 14.4 ms step (tools/tools_step_timeline.py,
        profiles/r03_step_timeline.txt).  Instead the tensors are kept until this object dies, then parked with ONE
        event recorded on the consumer stream and dropped when that event has passed (`_reap_handed_over`).
        Everything that enqueues reads of this geometry on `consumer` holds THIS object until it has enqueued them --
        the SparseConvNetTensors of the pass, the compiled `_Pass` (planExecutor: `ps.md`, kept by the autograd node
        until its backward list is out) and every per-layer Function ctx (`ctx.input_metadata`) -- so the parking event
        recorded in __del__ lies behind the last such read."""
        self._handed_over = (consumer, self.device_tensors())

    def __del__(self):
        ho = getattr(self, "_handed_over", None)
        if ho is None:
            return
        try:
            ev = torch.cuda.Event()
            ev.record(ho[0])
            with _park_lock:
                _parked.append((ev, ho[1]))
        except Exception:     # interpreter shutdown: the process's memory goes away with it
            pass

    def device_tensors(self):
        """every device tensor this object owns (for cross-stream hand-over of geometry prepared ahead of time)"""
        ts = []
        pend = getattr(self, "_pending", None)
        if pend is not None:
            ts += [pend["buf"]]
        for g in self.grids.values():
            ts += [g.coords] + ([g.keys] if g.keys is not None else []) + (g.brick.tensors() if g.brick is not None else [])
        ts += list(self._brick_keep)
        if self.input is not None:
            ts += [self.input["point_site"]]
        for tb in list(self.submanifold.values()) + list(self.rulebooks.values()):
            for ga in (tb.out, tb.inn):
                if ga is None:
                    continue
                ts += [t for t in (ga.table, ga.counts, ga._blocks, ga._pairs) if t is not None]
                if ga._blocks256:
                    ts += list(ga._blocks256.values())
        return ts

    def getNActive(self, spatial_size):
        if getattr(self, "_inb", None) is not None and _key(spatial_size) == self.input_spatial:
            return len(self._inb["locs"])
        return self.grids[_key(spatial_size)].V

    def setInputSpatialSize(self, spatial_size):
        """Metadata::setInputSpatialSize (Metadata.cpp:76-80)"""
        self.input_spatial = _key(spatial_size)

    # ---- incremental input construction (pybind.cpp:15-19; Metadata.cpp:81-145, 30-44) --------------------------
    # The reference fills the caller's `features` tensor on the HOST, one location at a time: a location not seen
    # before in the sample appends a row (first-seen numbering across all calls), a repeated one overwrites its row
    # or is ignored.  Same here, in Python (this is host bookkeeping in the reference too); the DEVICE grid of the
    # collected sites is built at the first geometry query (`_materialise_input`) through the voxel-scatter kernels.
    def batchAddSample(self):
        assert getattr(self, "input_spatial", None) is not None, "Call setInputSpatialSize first, please!"
        b = self.__dict__.setdefault("_inb", dict(rows={}, locs=[], nsamples=0))
        b["nsamples"] += 1

    def _add_point(self, features, key, vec, overwrite, rows_out):
        b = self._inb
        r = b["rows"].get(key)
        if r is None:
            b["rows"][key] = len(b["locs"])
            b["locs"].append(key)
            rows_out.append((len(b["locs"]) - 1, vec))
        elif overwrite:
            rows_out.append((r, vec))

    def _flush_rows(self, features, planes, rows):
        n = len(self._inb["locs"])
        old = features.clone() if features.numel() else None
        features.resize_(n, planes)
        if old is not None:
            features[: old.shape[0]] = old
        for r, vec in rows:
            features[r] = vec

    def setInputSpatialLocation(self, features, location, vec, overwrite):
        assert getattr(self, "_inb", None) and self._inb["nsamples"] > 0, "call batchAddSample first"
        rows = []
        key = (self._inb["nsamples"] - 1,) + tuple(int(v) for v in location.tolist())
        self._add_point(features, key, vec, overwrite, rows)
        self._flush_rows(features, int(vec.shape[0]), rows)

    def setInputSpatialLocations(self, features, locations, vecs, overwrite):
        assert getattr(self, "input_spatial", None) is not None, "Call setInputSpatialSize first, please!"
        b = self.__dict__.setdefault("_inb", dict(rows={}, locs=[], nsamples=0))
        L = locations.tolist()
        rows = []
        for i, loc in enumerate(L):
            if len(loc) == self.dimension:
                assert b["nsamples"] > 0, "call batchAddSample first"
                key = (b["nsamples"] - 1,) + tuple(int(v) for v in loc)
            else:                                       # 4th column = sample index; grows the batch as needed
                key = (int(loc[-1]),) + tuple(int(v) for v in loc[:-1])
                b["nsamples"] = max(b["nsamples"], key[0] + 1)
            self._add_point(features, key, vecs[i], overwrite, rows)
        self._flush_rows(features, int(vecs.shape[1]), rows)

    def _materialise_input(self, device):
        """device grid of the incrementally collected sites (unique by construction, sample-major like the
        reference's per-sample grids with their `ctr` offsets)"""
        b = self.__dict__.pop("_inb", None)
        if b is None or self.input is not None:
            return
        order = sorted(range(len(b["locs"])), key=lambda i: (b["locs"][i][0], i))   # samples contiguous, first-seen inside
        if order != list(range(len(order))):
            raise _hip.AabrError("incremental input: add the samples one after the other (rows of a sample must be "
                                 "contiguous, as Metadata::getSpatialLocations' ctr offsets assume)")
        coords = torch.tensor([list(k[1:]) + [k[0]] for k in b["locs"]], dtype=torch.int64).reshape(-1, 4)
        self.inputLayer(torch.LongTensor(list(self.input_spatial)), coords, b["nsamples"], 3, device)

    # ---- builders ---------------------------------------------------------------------------
    def inputLayer(self, spatial_size, coords, batch_size, mode, device):
        """Metadata::inputLayer (Metadata.cpp:405-417)"""
        if self.input is not None:
            src = self.input.get("coords_src")
            if src is not None and src[0] is coords and src[1] == coords._version and self.input["mode"] == int(mode):
                return self.inputLayerFinish()  # prepared ahead of time (InputLayer.prepare)
        self.inputLayerEnqueue(spatial_size, coords, mode, device, asynchronous=False)
        return self.inputLayerFinish()

    def inputLayerEnqueue(self, spatial_size, coords, mode, device, asynchronous=True):
        """enqueue the site-numbering kernels (+ an asynchronous read-back of (V, maxActive, err)
        when `asynchronous`); all integer state of the input layer lives in ONE device allocation"""
        assert coords.dim() == 2 and coords.size(1) in (3, 4)
        assert self.input is None and len(self.grids) == 0, "Metadata already holds an input layer"
        coords_src = (coords, coords._version)   # identity + version of the caller's tensor (kept alive here)
        lib = _hip.load()
        self.device = device
        coords = coords.to(device=device, dtype=torch.int64, non_blocking=True).contiguous()
        n, ncols = coords.shape
        if self.site_order == "brick" and brick_scatter and n > 0:
            return self._enqueue_brick_scatter(spatial_size, coords, coords_src, mode, device, asynchronous)
        cap = _hip.next_pow2(2 * n)
        n1 = max(n, 1)
        nst = int(lib.aabr_input_layer_status_words(n))
        # Which form of the insert (csrc/voxel_scatter.hip): 2 = LDS-binned, 1 = one atomic per point, 0 = generic (two
        # atomics per point; any coordinates).  The first two need (batch, x, y, z, point index) in one 64-bit word --
        # widths from the layer's spatial size and n -- and are only worth their extra launch from a few 10^4 points on;
        # a point that does not fit is reported through meta[5] and the generic form runs instead (inputLayerFinish).
        sp3 = _hip.i32x3(_key(spatial_size))
        variant = scatter_variant
        if variant and (n < SCATTER_MIN_POINTS or lib.aabr_input_layer_pack_bits(n, sp3) < 2):
            variant = 0
        if variant == 2 and not (4096 <= cap <= (1 << 24)):
            variant = 1
        # int32 words, 16-byte aligned pieces; the grid's 16-byte entries {key, first, val} and meta sit back to back
        # so the library clears them with ONE fill:  grid(4*cap) meta(8) | slot(n) point_site(n) nxt(n)
        #                      site_coords(4*n1) first_pt(n1) cnt_extra(n1) head(n1) last_pt(n1) status(nst)
        names = [("keys", 4 * cap), ("meta", _hip.META_WORDS), ("slot", n),
                 ("point_site", n), ("nxt", n), ("site_coords", 4 * n1), ("first_pt", n1), ("cnt_extra", n1),
                 ("head", n1), ("last_pt", n1), ("status", nst)]
        if variant:      # scratch of the packed forms: cap 8-byte words + one cursor per 4096-slot block
            names += [("words", 2 * cap), ("cursor", max(cap // 4096, 1))]
        offs, tot = {}, 0
        for name, sz in names:
            offs[name] = tot
            tot += (sz + 3) & ~3
        buf = torch.empty(tot, dtype=torch.int32, device=device)
        piece = {name: buf[offs[name]:offs[name] + sz] for name, sz in names}
        keys = piece["keys"].view(torch.int64)    # cap entries of 16 bytes: {uint64 key, uint32 first, int32 val}
        vals, meta = None, piece["meta"]
        site_coords = piece["site_coords"].view(n1, 4)
        base = buf.data_ptr()
        P = lambda name: base + 4 * offs[name]
        host = ev = None
        if n > 0:
            if variant:
                check(lib.aabr_input_layer_sites_packed(ptr(coords), n, ncols, sp3, variant, P("keys"), cap, P("words"),
                                                        P("cursor"), P("slot"), P("point_site"), P("site_coords"),
                                                        P("first_pt"), P("cnt_extra"), P("head"), P("nxt"), P("status"),
                                                        P("meta"), stream()))
            else:
                check(lib.aabr_input_layer_sites(ptr(coords), n, ncols, P("keys"), cap, P("slot"),
                                                 P("point_site"), P("site_coords"), P("first_pt"), P("cnt_extra"),
                                                 P("head"), P("nxt"), P("status"), P("meta"), stream()))
            if asynchronous:
                # pinned read-back buffer + event from a small recycling pool (allocating pinned memory
                # per scene costs more than the whole enqueue)
                host, ev = _pinned_pool.pop() if _pinned_pool else (
                    torch.empty(_hip.META_WORDS, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
                host.copy_(meta, non_blocking=True)
                ev.record()
        else:  # empty scene: an empty grid (all keys EMPTY), nothing to launch
            keys.fill_(-1)
            meta.zero_()
        self._pending = dict(host=host, event=ev, meta=meta, site_coords=site_coords, keys=keys, vals=vals,
                             cap=cap, coords=coords, buf=buf, variant=variant,
                             redo=(lambda: check(lib.aabr_input_layer_sites(
                                 ptr(coords), n, ncols, P("keys"), cap, P("slot"), P("point_site"), P("site_coords"),
                                 P("first_pt"), P("cnt_extra"), P("head"), P("nxt"), P("status"), P("meta"), stream()))))
        scatter_stats["variant%d" % variant] += 1
        self.input = dict(point_site=piece["point_site"], first_pt=piece["first_pt"], cnt_extra=piece["cnt_extra"],
                          head=piece["head"], nxt=piece["nxt"], last_pt=piece["last_pt"], meta=meta, n=n, V=None,
                          mode=int(mode), spatial=_key(spatial_size), coords_src=coords_src)

    def _enqueue_brick_scatter(self, spatial_size, coords, coords_src, mode, device, asynchronous):
        """brick-major rows WITHOUT a hash table (csrc/brick.hip "voxel scatter straight into a brick grid"): the points are
        converted and their extent reduced now (one kernel + the asynchronous read of the extent); `inputLayerFinish` sizes
        the directory by it, builds the input level from the points and hands every point its row"""
        lib = _hip.load()
        n, ncols = coords.shape
        names = [("pc", 4 * n), ("meta", _hip.META_WORDS), ("point_site", n), ("nxt", n), ("first_pt", n), ("head", n),
                 ("cnt_extra", n), ("last_pt", n)]
        offs, tot = {}, 0
        for name, sz in names:
            offs[name] = tot
            tot += (sz + 3) & ~3 if name != "first_pt" else sz          # first_pt | head contiguous: ONE fill for the two
        buf = torch.empty(tot + 4, dtype=torch.int32, device=device)
        piece = {name: buf[offs[name]:offs[name] + sz] for name, sz in names}
        meta = piece["meta"]
        # (the per-site arrays' starting values ride in this pass: aabr_points_sites(flags = 1) then fills nothing)
        check(lib.aabr_points_prepare(ptr(coords), n, ncols, piece["pc"].data_ptr(), meta.data_ptr(),
                                      ptr(piece["first_pt"]), ptr(piece["cnt_extra"]), ptr(piece["head"]), stream()))
        host = ev = None
        if asynchronous:
            host, ev = _pinned_pool.pop() if _pinned_pool else (
                torch.empty(_hip.META_WORDS, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
            host.copy_(meta, non_blocking=True)
            ev.record()
        self._pending = dict(kind="brick", host=host, event=ev, meta=meta, coords=coords, buf=buf, piece=piece,
                             spatial_t=spatial_size, mode=mode, device=device)
        scatter_stats["brick"] = scatter_stats.get("brick", 0) + 1
        self.input = dict(point_site=piece["point_site"], first_pt=piece["first_pt"], cnt_extra=piece["cnt_extra"],
                          head=piece["head"], nxt=piece["nxt"], last_pt=piece["last_pt"], meta=meta, n=n, V=None,
                          mode=int(mode), spatial=_key(spatial_size), coords_src=coords_src)

    def _brick_scatter_launch(self, piece, n, key, extent, dev):
        """the launches of the brick-native scatter behind the extent read: input level from the points, then every point's
        row / first point / chain (no host read in here -- bench.py times exactly this)"""
        lib = _hip.load()
        # the rows of every level's read-back block (sample offsets + meta) start at zero: ONE fill here covers the meta words
        # of the input level and of the pyramid built later (round 5: a memset per level build + a fill per pyramid)
        rows = self._brick_rows = torch.zeros((_BK_MAX_LEVELS, _BK_ROW), dtype=torch.int32, device=dev)
        self._brick_rows_clean = True
        self._brick_nrows = 1
        bk = _Brick([e + 1 for e in extent[:3]], extent[3] + 1, n, n, dev, rows[0, 64:64 + _hip.META_WORDS], alloc=False)
        # ONE allocation and ONE fill for the directory, the bricks and the scan's scratch (round 5: two memsets)
        lw = 4 * (bk.nw + bk.nb_cap)
        sw = (int(lib.aabr_brick_scratch_words(bk.nw, bk.nb_cap)) + 3) // 4 * 4
        arena = torch.zeros(lw + sw, dtype=torch.int32, device=dev)
        bk.level = arena[:lw]
        # the level is clamped to the EXTENT of the points, not to the layer's spatial size: a coordinate beyond
        # spatial_size (<= 65534) is a site here as it is for the hash form and the reference's InputLayer
        out_sp = tuple(max(int(k), int(e) + 1) for k, e in zip(key, extent[:3]))
        scratch = _brick_build(piece["pc"], n, 0, (1, 1, 1), (1, 1, 1), out_sp, bk, 1, arena[lw:])
        flush_geom()
        check(lib.aabr_points_sites(piece["pc"].data_ptr(), n, bk.dims_c(), bk.dir_ptr(), bk.bricks_ptr(),
                                    ptr(piece["point_site"]), ptr(piece["first_pt"]), ptr(piece["cnt_extra"]),
                                    ptr(piece["head"]), ptr(piece["nxt"]), bk.meta.data_ptr(), 1, stream()))
        return bk, scratch

    def _finish_brick_scatter(self, pend):
        lib = _hip.load()
        if pend["event"] is not None:
            pend["event"].synchronize()
            m = pend["host"].tolist()
            if len(_pinned_pool) < 16:
                _pinned_pool.append((pend["host"], pend["event"]))
        else:
            m = _hip.read_back(pend["meta"])
        self._pending = None
        il = self.input
        key, n, dev, piece = il["spatial"], il["n"], pend["device"], pend["piece"]
        if m[2] == 0:
            raise _hip.AabrError("InputLayer: coordinates must lie in [0, 65534] (batch index too)")
        if m[8] < 0:                                   # no valid point
            self.grids[key] = _Grid(torch.empty((0, 4), dtype=torch.int32, device=dev), None, None, 0, 0)
            il["V"] = 0
            self.input_spatial = key
            return
        if _brick_dir_words(m[8:12]) > BRICK_MAX_DIR_WORDS:
            # a few sites spread over a huge extent: the dense directory would not pay -- hash grids for this scene
            self.site_order = "first_seen"
            brick_stats["declined"] += 1
            src = il["coords_src"]
            self.input = None
            self.inputLayerEnqueue(pend["spatial_t"], pend["coords"], pend["mode"], dev, asynchronous=False)
            self.input["coords_src"] = src
            return self.inputLayerFinish()
        brick_stats["built"] += 1
        bk, scratch = self._brick_scatter_launch(piece, n, key, m[8:12], dev)
        bm = _hip.read_back(bk.meta)                   # the site count sizes every later tensor
        if bm[2]:
            raise _hip.AabrError("brick grid of the input level: a point outside the directory's extent or a capacity "
                                 "overflow (meta %s)" % (bm[:4],))
        V = bm[0]
        self._brick_keep = [pend["buf"], self._brick_rows, scratch]
        self.grids[key] = _Grid(bk.coords_full[:V], None, None, 0, V, None, bk)
        il["V"] = V
        self.input_spatial = key

    def inputLayerFinish(self):
        """wait for the read-back (the one host sync of the input layer: V sizes every later tensor)"""
        pend = getattr(self, "_pending", None)
        if pend is not None and pend.get("kind") == "brick":
            self._finish_brick_scatter(pend)
            return self.input["V"]
        if pend is not None:
            if pend["event"] is not None:
                pend["event"].synchronize()
                m = pend["host"].tolist()
                if len(_pinned_pool) < 16:
                    _pinned_pool.append((pend["host"], pend["event"]))
            else:
                m = pend["meta"].tolist()  # synchronous read-back
            if pend["variant"] and m[5] == 0:
                # a point did not fit the packed word (or a hash block overflowed): the generic form on the same buffers
                scatter_stats["redone"] += 1
                pend["redo"]()
                m = pend["meta"].tolist()
            self._pending = None
            if m[2]:
                raise _hip.AabrError("InputLayer: coordinates must lie in [0, 65534] (batch index too)")
            V = m[0]
            key = self.input["spatial"]
            if self.site_order == "brick" and V > 0 and _brick_dir_words(m[8:12]) > BRICK_MAX_DIR_WORDS:
                # a few sites spread over a huge extent: the dense directory would not pay -- hash grids for this scene
                self.site_order = "first_seen"
                brick_stats["declined"] += 1
            if self.site_order == "brick" and V > 0:
                brick_stats["built"] += 1
                self.grids[key] = self._brickify_input(pend, V, m[8:12])
            else:
                self.grids[key] = _Grid(pend["site_coords"][:V], pend["keys"], pend["vals"], pend["cap"], V)
            self.input["V"] = V   # maxActive (meta[1]) is produced by the forward kernel: read lazily (max_active())
            self.input_spatial = key
        return self.input["V"]

    # ---- brick-major site order (extension; csrc/brick.hip) -------------------------------------------------------------
    def _brickify_input(self, pend, V, extent):
        """The voxel scatter has numbered the V input sites in first-seen order and the host knows V and the scene's extent
        (meta[8..11]).  Build the input level's brick grid from those sites, renumber them brick by brick and carry the
        input layer's per-site arrays (first point, further-point chain, point -> site) over to the new rows.  Nothing is
        read back here; the level's error flag rides with the next read (`buildBrickPyramid` / `getRuleBook`)."""
        lib = _hip.load()
        il = self.input
        dev = pend["buf"].device
        n = il["n"]
        ext = [e + 1 for e in extent[:3]]
        nsamples = extent[3] + 1
        rows = self._brick_rows = torch.empty((_BK_MAX_LEVELS, _BK_ROW), dtype=torch.int32, device=dev)
        self._brick_nrows = 1
        bk = _Brick(ext, nsamples, V, V, dev, rows[0, 64:64 + _hip.META_WORDS])
        old_coords = pend["site_coords"]
        out_sp = tuple(max(int(k), int(e)) for k, e in zip(il["spatial"], ext))       # see _brick_scatter_launch
        scratch = _brick_build(old_coords, V, 0, (1, 1, 1), (1, 1, 1), out_sp, bk)
        flush_geom()
        nb = torch.empty(5 * V + n + 8, dtype=torch.int32, device=dev)
        new_of_old, old_of_new, first2, extra2, head2 = (nb[i * V:(i + 1) * V] for i in range(5))
        ps2 = nb[5 * V:5 * V + n]
        check(lib.aabr_brick_renumber(ptr(old_coords), V, bk.dims_c(), bk.dir_ptr(), bk.bricks_ptr(), ptr(new_of_old),
                                      ptr(old_of_new), ptr(il["first_pt"]), ptr(il["cnt_extra"]), ptr(il["head"]),
                                      ptr(first2), ptr(extra2), ptr(head2), ptr(il["point_site"]), n, ptr(ps2),
                                      bk.meta.data_ptr(), stream()))
        il.update(point_site=ps2, first_pt=first2, cnt_extra=extra2, head=head2, new_of_old=new_of_old,
                  old_of_new=old_of_new)
        self._brick_keep = [nb, rows, scratch]
        g = _Grid(bk.coords_full[:V], None, None, 0, V, None, bk)
        return g

    def _brick_read(self, post_only=False, posted=None):
        """ONE host read for every brick level enqueued since the last one: [V, per-sample offsets (64 words), meta (16)]
        per level; checks the error flags.  `post_only`: enqueue the read (a mailbox of this thread's own) and return the
        handle; `posted`: collect such a read -- the host does other work while the levels are built."""
        n = self._brick_nrows
        if posted is None:
            flush_geom()
            if post_only:
                mb = _geom_mailboxes.pop() if _geom_mailboxes else _hip.Mailbox()
                mb.post(self._brick_rows[:n].reshape(-1))
                return (mb, n)
            rows = _hip.read_back(self._brick_rows[:n].reshape(-1))
        else:
            mb, n = posted
            rows = mb.wait()
            if len(_geom_mailboxes) < 4:
                _geom_mailboxes.append(mb)
        rows = [rows[i * _BK_ROW:(i + 1) * _BK_ROW] for i in range(n)]
        for i, r in enumerate(rows):
            if r[64 + 2]:
                raise _hip.AabrError("brick grid %d: a site outside the directory's extent or a capacity overflow "
                                     "(meta %s)" % (i, r[64:64 + 4]))
        g0 = self.grids[self.input["spatial"]]
        if rows[0][64] != g0.V:
            raise _hip.AabrError("brick grid of the input level holds %d sites, the voxel scatter counted %d"
                                 % (rows[0][64], g0.V))
        return rows

    def finishBrickPyramid(self):
        """collect a pyramid enqueued with `buildBrickPyramid(.., defer=True)` (no-op when there is none)"""
        pend = self.__dict__.pop("_pyramid_pending", None)
        if pend is not None:
            self._pyramid_collect(*pend)

    def buildBrickPyramid(self, specs, defer=False):
        """Extension: the strided levels of a pass as brick grids, each built from its parent LEVEL by device-side counts
        (aabr_brick_build), all enqueued back to back, ONE host read for all of their site counts and per-sample offsets.
        specs = [(out_spatial key, source spatial key, size, stride)] in dependency order."""
        self.finishBrickPyramid()
        todo = [sp for sp in specs if sp[0] not in self.grids]
        if not todo:
            return
        dev = self._brick_rows.device
        lib = _hip.load()
        made = {}
        first_row = self._brick_nrows
        plan = []
        for osz, src, size, stride in todo:
            gs = made.get(src) or self.grids[src]
            bs = gs.brick
            ext = [min(o, (e - 1) // st_ + 1) for o, e, st_ in zip(osz, bs.ext, stride)]
            maxout = 1
            for a, b in zip(size, stride):
                maxout *= (a + b - 1) // b
            row = self._brick_rows[self._brick_nrows]
            self._brick_nrows += 1
            assert self._brick_nrows <= _BK_MAX_LEVELS
            bk = _Brick(ext, bs.dims[3], bs.nb_cap * maxout, bs.v_cap * maxout, dev, row[64:64 + _hip.META_WORDS], alloc=False)
            made[osz] = _Grid(None, None, None, 0, None, None, bk)
            plan.append((osz, bs, size, stride, bk, row))
        # ONE allocation and ONE fill for the directories, bricks and scan scratch of all the levels built here
        words, offs = 0, []
        for osz, bs, size, stride, bk, row in plan:
            lw = 4 * (bk.nw + bk.nb_cap)
            sw = (int(lib.aabr_brick_scratch_words(bk.nw, bk.nb_cap)) + 3) // 4 * 4
            offs.append((words, lw, sw))
            words += lw + sw
        arena = torch.zeros(max(words, 4), dtype=torch.int32, device=dev)
        if not self.__dict__.get("_brick_rows_clean"):      # (rows allocated zeroed by the brick-native scatter: nothing to do)
            self._brick_rows[first_row:self._brick_nrows].zero_()
        self._brick_keep.append(arena)
        for (osz, bs, size, stride, bk, row), (o, lw, sw) in zip(plan, offs):
            bk.level = arena[o:o + lw]
            _brick_build(bs.coords_full, bs.v_cap, bs.meta.data_ptr(), size, stride, osz, bk, 1, arena[o + lw:o + lw + sw])
            _geom(G_SOFF, (MAX_SAMPLES + 1,), (bk.v_cap,), (_p(bk.coords_full), bk.meta.data_ptr(), row.data_ptr()))
        if defer:        # the levels are being built; the host collects their counts later (finishBrickPyramid)
            self._pyramid_pending = (todo, made, first_row, self._brick_read(post_only=True))
            return
        self._pyramid_collect(todo, made, first_row, None)

    def _pyramid_collect(self, todo, made, first_row, posted):
        rows = self._brick_read(posted=posted) if posted is not None else self._brick_read()
        pre = self.__dict__.setdefault("_pregrids", set())
        for (osz, src, size, stride), r in zip(todo, rows[first_row:]):
            g = made[osz]
            V = r[0]
            off = r[1:64]
            g.V, g.coords = V, g.brick.coords_full[:V]
            g.sample_off = off if off[-1] == V else None
            self.grids[osz] = g
            pre.add(osz)

    def getSubmanifoldRuleBook(self, spatial_size, filter_size):
        """Metadata::getSubmanifoldRuleBook (Metadata.cpp:429-443) -> cached gather table"""
        k = _key(spatial_size) + _key(filter_size)
        tb = self.submanifold.get(k)
        if tb is None:
            if getattr(self, "_inb", None) is not None:
                self._materialise_input(torch.device("cuda", torch.cuda.current_device()))
            g = self.grids[_key(spatial_size)]
            fs = _key(filter_size)
            vol = fs[0] * fs[1] * fs[2]
            dev = g.coords.device
            table = torch.empty((vol, g.V), dtype=torch.int32, device=dev)
            counts = torch.empty(vol * ((g.V + 255) // 256), dtype=torch.int32, device=dev)
            if g.V > 0 and g.brick is not None:
                _geom(G_BK_SUBM, fs, (g.V, g.brick.packed), (_p(g.coords), g.brick.level.data_ptr(), _p(table), _p(counts)))
            elif g.V > 0:
                _geom(G_SUBM, fs, (g.V, g.cap), (_p(g.coords), _p(g.keys), _p(table), _p(counts)))
            # odd filters: the input-gradient gather is the same table read with the mirrored
            # offset (u = v + off_k  <=>  v = u + off_{vol-1-k})
            tb = _Table(_Gather(table, counts, vol, g.V), None, vol, g.V, g.V,
                        flip_ok=all(f % 2 == 1 for f in fs))
            if not tb.flip_ok:
                raise NotImplementedError("even-sized submanifold filters")
            self.submanifold[k] = tb
        return tb

    def buildGridsFromInput(self, in_spatial, specs):
        """Extension: the output grids of a chain of NON-OVERLAPPING strided levels (filter size == stride at every
        level) built straight from the chain's first grid -- `specs` = [(out_spatial, composed)], composed = the
        per-axis product of the strides down to that level -- all enqueued back to back and sized by ONE read of
        their site counts (level by level every grid costs its own blocking read: 12 per FPN_Net pass).  A site of
        level k is (x >> ..) of its level-0 sites and is numbered at its first level-0 site, which is also where the
        level-by-level construction numbers it: same site lists, same order (tests compare them with the oracle's).
        `getRuleBook` picks the grids up by their spatial size."""
        lib = _hip.load()
        gi = self.grids[_key(in_spatial)]
        if gi.brick is not None:       # brick grids: each level from the round's base with the composed stride
            return self.buildBrickPyramid([(_key(o), _key(in_spatial), _key(c), _key(c)) for o, c in specs])
        dev = gi.keys.device
        E = gi.V
        cap = derived_cap(E)
        nblk = (max(E, 1) + 255) // 256
        metas = torch.empty((max(len(specs), 1), _hip.META_WORDS), dtype=torch.int32, device=dev)
        ext = torch.empty((max(len(specs), 1), MAX_SAMPLES + 3), dtype=torch.int32, device=dev)
        pend = []
        for i, (osz, comp) in enumerate(specs):
            osz, comp = _key(osz), _key(comp)
            keys = torch.empty(2 * cap, dtype=torch.int64, device=dev)   # cap 16-byte entries {key, first, val}
            vals = None
            scratch = torch.empty(E + 4 * nblk + 16, dtype=torch.int32, device=dev)
            out_coords = torch.empty((max(E, 1), 4), dtype=torch.int32, device=dev)
            _geom(G_SITES, comp + comp + osz, (gi.V, cap),
                  (_p(gi.coords), _p(keys), _p(scratch), _p(out_coords), metas[i].data_ptr()))
            # V and the per-sample row offsets (SparseGrid::ctr) of the new grid into one row of `ext`
            _geom(G_SOFF, (MAX_SAMPLES + 1,), (max(E, 1),), (_p(out_coords), metas[i].data_ptr(), ext[i].data_ptr()))
            _keep(scratch)
            pend.append((osz, out_coords, keys, vals))
        if not pend:
            return
        _keep(metas)
        flush_geom()
        rows = ext[:len(pend)].tolist()  # the one host sync: site counts + per-sample offsets of every grid
        pre = self.__dict__.setdefault("_pregrids", set())
        for (osz, out_coords, keys, vals), row in zip(pend, rows):
            V_out = row[0]
            off = row[1:]
            self.grids[osz] = _Grid(out_coords[:V_out], keys, vals, cap, V_out,
                                    off if off[-1] == V_out else None)   # more than MAX_SAMPLES samples: no list
            pre.add(osz)

    def getRuleBook(self, in_spatial, out_spatial, filter_size, filter_stride):
        """Metadata::getRuleBook (Metadata.cpp:484-510): builds the output grid on first use."""
        k = _key(in_spatial) + _key(filter_size) + _key(filter_stride)
        tb = self.rulebooks.get(k)
        if tb is None:
            lib = _hip.load()
            if getattr(self, "_inb", None) is not None:
                self._materialise_input(torch.device("cuda", torch.cuda.current_device()))
            gi = self.grids[_key(in_spatial)]
            fs, st, osz = _key(filter_size), _key(filter_stride), _key(out_spatial)
            dev = gi.coords.device
            vol = fs[0] * fs[1] * fs[2]
            have = self.grids.get(osz)
            if gi.brick is not None and gi.V > 0 and osz not in self.__dict__.get("_pregrids", ()) and \
                    not (have is not None and have.brick is not None):
                self.buildBrickPyramid([(osz, _key(in_spatial), fs, st)])       # (one host read)
            elif self.site_order == "brick" and gi.V == 0:                      # an empty level under an empty level
                self.grids[osz] = _Grid(gi.coords[:0], None, None, 0, 0)
                e = lambda r: torch.empty((vol, 0), dtype=torch.int32, device=dev)
                tb = _Table(_Gather(e(0), torch.empty(0, dtype=torch.int32, device=dev), vol, 0),
                            _Gather(e(0), torch.empty(0, dtype=torch.int32, device=dev), vol, 0), vol, 0, 0)
                self.rulebooks[k] = tb
                return tb
            go = self.grids.get(osz) if osz in self.__dict__.get("_pregrids", ()) else None
            if go is None and gi.brick is not None:
                # brick order: an output level that already exists (a second rule book onto the same output size, with
                # another filter or stride) is served from that level, as the reference serves it from its existing grid
                # (Metadata.cpp:484-510) -- rebuilding it would renumber rows that already carry features.  A brick
                # level never falls through to the hash builders below.
                go = self.grids.get(osz)
                if go is None or go.brick is None:
                    raise RuntimeError("site_order='brick': no brick level for output size %r" % (osz,))
            if go is not None:         # built ahead by buildGridsFromInput / buildBrickPyramid
                self.__dict__.setdefault("_pregrids", set()).discard(osz)
                V_out = go.V
                t_out = torch.empty((vol, V_out), dtype=torch.int32, device=dev)
                t_in = torch.empty((vol, gi.V), dtype=torch.int32, device=dev)
                counts = torch.empty(vol * ((V_out + 255) // 256), dtype=torch.int32, device=dev)
                counts_in = torch.empty(vol * ((gi.V + 255) // 256), dtype=torch.int32, device=dev)
                if gi.brick is not None:
                    _geom(G_BK_TABLES, fs + st + osz, (gi.V, V_out, gi.brick.packed, go.brick.packed),
                          (_p(gi.coords), gi.brick.level.data_ptr(), _p(go.coords), go.brick.level.data_ptr(), _p(t_out),
                           _p(t_in), _p(counts), _p(counts_in)))
                else:
                    _geom(G_TABLES, fs + st + osz, (gi.V, gi.cap, V_out, go.cap),
                          (_p(gi.coords), _p(gi.keys), _p(go.coords), _p(go.keys), _p(t_out), _p(t_in), _p(counts),
                           _p(counts_in)))
                tb = _Table(_Gather(t_out, counts, vol, V_out), _Gather(t_in, counts_in, vol, gi.V), vol, V_out, gi.V)
                self.rulebooks[k] = tb
                return tb
            flush_geom()           # the calls below launch (and read) right away
            maxout = 1
            for a, b in zip(fs, st):
                maxout *= (a + b - 1) // b
            E = gi.V * maxout
            cap = derived_cap(E)
            nblk = (max(E, 1) + 255) // 256
            keys = torch.empty(2 * cap, dtype=torch.int64, device=dev)   # cap 16-byte entries {key, first, val}
            vals = None
            scratch = torch.empty(E + 4 * nblk + 16, dtype=torch.int32, device=dev)
            out_coords = torch.empty((max(E, 1), 4), dtype=torch.int32, device=dev)
            meta = torch.empty(_hip.META_WORDS, dtype=torch.int32, device=dev)
            check(lib.aabr_convolution_sites(ptr(gi.coords), gi.V, _hip.i32x3(fs), _hip.i32x3(st),
                                             _hip.i32x3(osz), ptr(keys), cap, ptr(scratch),
                                             ptr(out_coords), ptr(meta), stream()))
            V_out = meta.tolist()[0]  # host sync: sizes the output feature tensor
            go = _Grid(out_coords[:V_out], keys, vals, cap, V_out)
            self.grids[osz] = go
            t_out = torch.empty((vol, V_out), dtype=torch.int32, device=dev)
            t_in = torch.empty((vol, gi.V), dtype=torch.int32, device=dev)
            counts = torch.empty(vol * ((V_out + 255) // 256), dtype=torch.int32, device=dev)
            counts_in = torch.empty(vol * ((gi.V + 255) // 256), dtype=torch.int32, device=dev)
            check(lib.aabr_convolution_tables2(ptr(gi.coords), gi.V, ptr(gi.keys), gi.cap,
                                               ptr(go.coords), V_out, ptr(go.keys), go.cap,
                                               _hip.i32x3(fs), _hip.i32x3(st), _hip.i32x3(osz), ptr(t_out),
                                               ptr(t_in), ptr(counts), ptr(counts_in), stream()))
            tb = _Table(_Gather(t_out, counts, vol, V_out), _Gather(t_in, counts_in, vol, gi.V), vol, V_out, gi.V)
            self.rulebooks[k] = tb
        return tb

    # ---- reference-format views (tests / API parity) ------------------------------------------
    def max_active(self):
        """largest number of points in one voxel (IOLayersRules.h:96-103); written by the forward kernel"""
        il = self.input
        if il.get("max_active") is None:
            if not il.get("forward_done"):
                raise _hip.AabrError("maxActive is known after InputLayer_updateOutput has run")
            il["max_active"] = max(int(il["meta"][1].item()), 0) if il["V"] else 0
        return il["max_active"]

    def inputLayerRuleBook(self):
        """[[mode, maxActive, nIn, nOut], rules V x (1+maxActive)] (IOLayersRules.h:10-15)"""
        il = self.input
        ma = self.max_active()
        w = (ma if il["mode"] in (3, 4) else 1) + 1
        rules = torch.empty((il["V"], w), dtype=torch.int32, device=il["first_pt"].device)
        check(_hip.load().aabr_input_layer_rule_table(ptr(il["first_pt"]), ptr(il["last_pt"]), ptr(il["cnt_extra"]),
                                                      ptr(il["head"]), ptr(il["nxt"]), il["V"], ma, il["mode"],
                                                      ptr(rules), stream()))
        return [il["mode"], w - 1, il["n"], il["V"]], rules

    @staticmethod
    def tableToRuleBook(table):
        """gather table [vol, V] -> list over offsets of int32 [n_k, 2] (entry, row) pairs"""
        vol, V = table.shape
        rules = torch.empty((vol, max(V, 1), 2), dtype=torch.int32, device=table.device)
        counts = torch.empty(vol, dtype=torch.int32, device=table.device)
        check(_hip.load().aabr_table_to_rulebook(ptr(table), V, vol, ptr(rules), ptr(counts), stream()))
        c = counts.tolist()
        return [rules[k, : c[k]] for k in range(vol)]


def _opt(t):
    return t if (t is not None and t.numel() > 0) else None


def _featc(t, what):
    """feature matrices of the convolution / batch-norm path: fp32 (the reference's precision) or
    bf16 storage (extension, BASELINE configs 3-5; fp32 accumulation, fp32 parameters)"""
    _hip.require_gpu(t)
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("%s must be float32 or bfloat16, got %s" % (what, t.dtype))
    return t.contiguous()


def _same_dtype(a, b, what):
    if a.dtype != b.dtype:
        raise TypeError("%s: feature dtypes differ (%s vs %s)" % (what, a.dtype, b.dtype))


def _f32c(t, what):
    _hip.require_gpu(t)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (the reference path is fp32-only, "
                        "sparseconvnet_cuda.cpp instantiates <float>), got %s" % (what, t.dtype))
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# InputLayer (pybind.cpp:154-162; sparseconvnet_cuda.cpp:432-455)
# ------------------------------------------------------------------------------------------------
def InputLayer_updateOutput(metadata, spatial_size, coords, input_features, output_features, batch_size,
                            mode):
    inp = _f32c(input_features, "input_features")
    mode = int(mode)
    if mode == 0:
        # "guaranteed unique": identical to the summing path when the guarantee holds
        V = metadata.inputLayer(spatial_size, coords, batch_size, 3, inp.device)
        if V != inp.size(0):
            raise _hip.AabrError("InputLayer mode 0 requires unique coordinates")
        metadata.input["mode"] = 3
    else:
        if mode not in (1, 2, 3, 4):
            raise ValueError("InputLayer mode must be 0..4")
        V = metadata.inputLayer(spatial_size, coords, batch_size, mode, inp.device)
    il = metadata.input
    planes = inp.size(1)
    output_features.resize_(V, planes)
    check(_hip.load().aabr_input_layer_forward(ptr(inp), ptr(output_features), V, planes, ptr(il["first_pt"]),
                                               ptr(il["cnt_extra"]), ptr(il["head"]), ptr(il["nxt"]),
                                               ptr(il["last_pt"]), il["mode"], ptr(il["meta"]), stream()))
    il["forward_done"] = True


def InputLayer_updateGradInput(metadata, d_input_features, d_output_features):
    d_out = _f32c(d_output_features, "d_output_features")
    il = metadata.input
    planes = d_out.size(1)
    d_input_features.resize_(il["n"], planes)
    check(_hip.load().aabr_input_layer_backward(ptr(d_input_features), ptr(d_out), il["n"], planes,
                                                ptr(il["point_site"]), ptr(il["first_pt"]), ptr(il["last_pt"]),
                                                ptr(il["cnt_extra"]), il["mode"], stream()))


# OutputLayer (pybind.cpp:163-170; SCN/CPU/IOLayers.cpp:95-134): the InputLayer run backwards without averaging --
# every point listed in the input rule table receives its site's row (cpu_OutputLayer_updateOutput calls
# InputLayer_BackwardPass(..., average=false)); its gradient is InputLayer_ForwardPass(..., average=false).
def OutputLayer_updateOutput(metadata, input_features, output_features):
    inp = _f32c(input_features, "input_features")
    il = metadata.input
    planes = inp.size(1)
    mode = 3 if il["mode"] == 4 else il["mode"]
    output_features.resize_(il["n"], planes)
    check(_hip.load().aabr_input_layer_backward(ptr(output_features), ptr(inp), il["n"], planes,
                                                ptr(il["point_site"]), ptr(il["first_pt"]), ptr(il["last_pt"]),
                                                ptr(il["cnt_extra"]), mode, stream()))


def OutputLayer_updateGradInput(metadata, d_input_features, d_output_features):
    d_out = _f32c(d_output_features, "d_output_features")
    il = metadata.input
    planes = d_out.size(1)
    mode = 3 if il["mode"] == 4 else il["mode"]
    d_input_features.resize_(il["V"], planes)
    check(_hip.load().aabr_input_layer_forward(ptr(d_out), ptr(d_input_features), il["V"], planes, ptr(il["first_pt"]),
                                               ptr(il["cnt_extra"]), ptr(il["head"]), ptr(il["nxt"]), None, mode,
                                               ptr(il["meta"]), stream()))


# ------------------------------------------------------------------------------------------------
# the shared gather-GEMM
# ------------------------------------------------------------------------------------------------
def _conv_fwd(inp, out, n_rows_out, gather, weight, bias, flags, pack_t=None):
    """`pack_t` (extension): a list owned by the autograd node.  The forward call (flags bit0 = 0) packs
    the weights in both orientations with one launch and leaves the input-gradient layout in it; the
    backward call (bit0 = 1) finds it there and runs without a pack launch of its own."""
    lib = _hip.load()
    if weight.size(1) != 1:
        return _conv_fwd_groups(inp, out, n_rows_out, gather, weight, bias, flags)
    n_in = inp.size(1)
    w = weight.contiguous()
    if flags & 1:
        n_out = w.size(2)
        assert w.size(3) == n_in
    else:
        n_out = w.size(3)
        assert w.size(2) == n_in, (w.shape, n_in)
    assert gather.rows == n_rows_out
    _same_dtype(inp, out, "convolution")
    out.resize_(n_rows_out, n_out)
    if w.dtype != torch.float32:
        raise TypeError("convolution weights are fp32 master parameters, got %s" % w.dtype)
    bf16 = inp.dtype == torch.bfloat16
    if narrow_ok(n_in, n_out, inp.size(0), n_rows_out, gather.vol, bf16):
        # 32 -> 32 planes: all offsets' weights in LDS, 16 output rows per wave in registers, straight from the gather
        # table (csrc/conv_narrow.hip) -- no weight pack, no block stream
        fn = lib.aabr_conv_forward_narrow_bf16 if bf16 else lib.aabr_conv_forward_narrow
        check(fn(ptr(inp), inp.size(0), ptr(out), n_rows_out, ptr(gather.table), gather.vol, ptr(w), ptr(_opt(bias)),
                 flags & 3, stream()))
        if trace is not None:
            trace.append(("fwd", n_in, n_out, gather, inp.size(0), flags & 3, inp.dtype))
        return n_out
    if bf16:
        elems, name, dt = lib.aabr_conv_wpack_bf16_elems(gather.vol, w.size(2), w.size(3)), "wpack16", torch.bfloat16
        conv, pack2 = lib.aabr_conv_forward_bf16, lib.aabr_conv_pack_weights2_bf16
    else:
        elems, name, dt = lib.aabr_conv_wpack_floats(gather.vol, w.size(2), w.size(3)), "wpack", torch.float32
        conv, pack2 = lib.aabr_conv_forward, lib.aabr_conv_pack_weights2
    pre = getattr(weight, "_aabr_pack", None)            # WeightPackPlan: packed once per weight version
    if pre is not None and pre[0] == weight._version and pre[1] == dt and n_rows_out > 0:
        wpack = pre[3 if (flags & 1) else 2]
        if pack_t is not None and not (flags & 1):           # the node's backward pass finds its layout here
            del pack_t[:]
            pack_t.append(pre[3])
        flags |= 4
        pack_stats["plan"] += 1
    elif pack_t is not None and (flags & 1) and len(pack_t) == 1 and pack_t[0].dtype == dt:
        wpack, flags = pack_t[0], flags | 4
    else:
        wpack = _hip.workspace(name, elems, dt, inp.device)
        if pack_t is not None and not (flags & 1) and n_rows_out > 0:
            wt = torch.empty(elems, dtype=dt, device=inp.device)
            check(pack2(ptr(w), gather.vol, w.size(2), w.size(3), ptr(wpack), ptr(wt), stream()))
            del pack_t[:]
            pack_t.append(wt)
            flags |= 4
    if not (flags & 4) and n_rows_out > 0:
        pack_stats["own"] += 1                               # this call packs its weights itself
    tile_rows = wide_tile_rows(n_in, n_out, inp.size(0), n_rows_out, gather.vol, bf16, bool(flags & 4))
    if tile_rows and bf16:
        check(lib.aabr_conv_forward_wide_bf16(ptr(inp), n_in, inp.size(0), ptr(out), n_out, n_rows_out,
                                              ptr(gather.blocks_wide(tile_rows)), tile_rows, gather.vol,
                                              ptr(_opt(bias)), flags & 3, ptr(wpack), stream()))
    elif tile_rows:
        # wide layer: big tiles, weights shared per offset (csrc/conv_wide.hip)
        if not (flags & 4):   # the LAUNCH's (n_in, n_out): for a transposed launch these are (w.size(3), w.size(2))
            check(lib.aabr_conv_pack_weights(ptr(w), gather.vol, n_in, n_out, flags & 1, ptr(wpack), stream()))
        check(lib.aabr_conv_forward_wide(ptr(inp), n_in, inp.size(0), ptr(out), n_out, n_rows_out,
                                         ptr(gather.blocks_wide(tile_rows)), tile_rows, gather.vol, ptr(_opt(bias)),
                                         flags & 3, ptr(wpack), stream()))
    elif (not bf16 or (flags & 4)) and wide_split(n_in, n_out, inp.size(0), n_rows_out, gather.vol, bf16):
        T, P = wide_split(n_in, n_out, inp.size(0), n_rows_out, gather.vol, bf16)
        scratch = _hip.workspace("wide_split", P * n_rows_out * n_out, torch.float32, inp.device)
        if bf16:
            check(lib.aabr_conv_forward_wide_split_bf16(ptr(inp), n_in, inp.size(0), ptr(out), n_out, n_rows_out,
                                                        ptr(gather.blocks_wide(T)), T, gather.vol, ptr(_opt(bias)),
                                                        flags & 3, ptr(wpack), P, ptr(scratch), stream()))
        else:
            if not (flags & 4):
                check(lib.aabr_conv_pack_weights(ptr(w), gather.vol, n_in, n_out, flags & 1, ptr(wpack), stream()))
            check(lib.aabr_conv_forward_wide_split(ptr(inp), n_in, inp.size(0), ptr(out), n_out, n_rows_out,
                                                   ptr(gather.blocks_wide(T)), T, gather.vol, ptr(_opt(bias)), flags & 3,
                                                   ptr(wpack), None, P, ptr(scratch), stream()))
    else:
        check(conv(ptr(inp), n_in, inp.size(0), ptr(out), n_out, n_rows_out, ptr(gather.blocks()), gather.vol,
                   ptr(w), ptr(_opt(bias)), flags, ptr(wpack), stream()))
    if trace is not None:
        trace.append(("fwd", n_in, n_out, gather, inp.size(0), flags & 3, inp.dtype))
    return n_out


def _conv_fwd_groups(inp, out, n_rows_out, gather, weight, bias, flags):
    """groups > 1 (the reference carries `groups` through every kernel: weight [vol, groups, nIn/g, nOut/g], planes
    group-major, out[:, g] = sum_k in[rule, g] @ W[k, g] -- CPU/Convolution.cpp:8-43,139-147).  Not on FPN_Net's path:
    one launch per group on contiguous copies of the group's planes, same kernels, no extra native code."""
    G = weight.size(1)
    ci, co = (weight.size(3), weight.size(2)) if flags & 1 else (weight.size(2), weight.size(3))
    assert inp.size(1) == G * ci, (inp.shape, weight.shape)
    out.resize_(n_rows_out, G * co)
    for g in range(G):
        xg = inp[:, g * ci:(g + 1) * ci].contiguous()
        og = torch.empty((n_rows_out, co), dtype=inp.dtype, device=inp.device)
        bg = bias[g * co:(g + 1) * co].contiguous() if (bias is not None and bias.numel()) else None
        _conv_fwd(xg, og, n_rows_out, gather, weight[:, g:g + 1].contiguous(), bg, flags & 3)
        out[:, g * co:(g + 1) * co] = og
    return G * co


pack_stats = {"plan": 0, "own": 0}   # convolution launches served by a WeightPackPlan / packing on their own


class WeightPackPlan:
    """Extension (the reference's GEMMs read W[k] in place): the packed copies of EVERY convolution weight of a
    network -- forward layout and input-gradient layout -- produced by ONE launch per forward pass
    (`aabr_conv_pack_weights_jobs`) instead of one pack launch per layer call.  `refresh()` packs and leaves each
    weight's packs on the parameter (`_aabr_pack`) for `_conv_fwd` to pick up; `release()` takes them away again
    when the forward pass is over (the autograd nodes keep the input-gradient packs they need), so a layer called
    on its own later -- possibly after an optimizer changed the weights through `.data`, which no version counter
    sees -- never meets a stale pack."""

    def __init__(self, weights, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.weights = [w for w in weights if w.is_cuda]
        self.dtype = dtype
        self._ptrs = None

    def _build(self):
        import struct
        lib = _hip.load()
        dev = self.weights[0].device
        es = 4 if self.dtype == torch.float32 else 2
        offs, total, blocks = [], 0, [0]
        for w in self.weights:
            assert w.dim() == 4 and w.size(1) == 1 and w.dtype == torch.float32 and w.is_contiguous()
            e = int(lib.aabr_conv_wpack_floats(w.size(0), w.size(2), w.size(3)))
            e = (e + 63) // 64 * 64                              # every pack 16-byte aligned in the arena
            offs.append((total, total + e, e))
            total += 2 * e
            blocks.append(blocks[-1] + int(lib.aabr_conv_pack_job_blocks(w.size(0), w.size(2), w.size(3))))
        self.arena = torch.empty(total, dtype=self.dtype, device=dev)
        base = self.arena.data_ptr()
        rec = b""
        self.packs = []
        for w, (f, t, e), b in zip(self.weights, offs, blocks):
            rec += struct.pack("<QQQiiiiq", w.data_ptr(), base + f * es, base + t * es, w.size(0), w.size(2), w.size(3),
                               0 if self.dtype == torch.float32 else 1, b)
            self.packs.append((self.arena[f:f + e], self.arena[t:t + e]))
        self.jobs = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
        self.total_blocks = blocks[-1]
        self._ptrs = [w.data_ptr() for w in self.weights]

    def refresh(self):
        if not self.weights:
            return
        if [w.data_ptr() for w in self.weights] != self._ptrs:
            self._build()
        check(_hip.load().aabr_conv_pack_weights_jobs(ptr(self.jobs), len(self.weights), self.total_blocks, stream()))
        for w, (pf, pt) in zip(self.weights, self.packs):
            w._aabr_pack = (w._version, self.dtype, pf, pt)

    def release(self):
        for w in self.weights:
            if hasattr(w, "_aabr_pack"):
                del w._aabr_pack


def wide_tile_rows(n_in, n_out, rows_in, rows_out, vol, bf16=False, prepacked=True):
    """rows per tile when this launch goes to the wide-layer kernel (csrc/conv_wide.hip), else 0 -- the one place the
    layer code, the stream pre-builder and the graph executor take that decision from.  bf16 storage: only with a
    prepacked weight (the training path always has one; a stand-alone call packs inside the 64-row-tile entry)."""
    if rows_out == 0:
        return 0
    lib = _hip.load()
    if bf16:
        return lib.aabr_conv_wide_tile_rows_bf16(n_in, n_out, rows_in, rows_out, vol) if prepacked else 0
    return lib.aabr_conv_wide_tile_rows(n_in, n_out, rows_in, rows_out, vol)


def narrow_ok(n_in, n_out, rows_in, rows_out, vol, bf16):
    """True when this launch goes to the 32 -> 32 kernel (csrc/conv_narrow.hip); asked FIRST by the layer code, the stream
    pre-builder and the graph executor alike"""
    if rows_out == 0:
        return False
    return bool(_hip.load().aabr_conv_narrow_ok(n_in, n_out, rows_in, rows_out, vol, 1 if bf16 else 0))


def wide_split(n_in, n_out, rows_in, rows_out, vol, bf16=False):
    """(tile_rows, parts) when this launch goes to the offset-split form of the wide kernel (coarse maps: too few
    (tile, slab) items to fill the chip; csrc/conv_wide.hip aabr_conv_forward_wide_split), else None.  Asked after
    `wide_tile_rows` declined; bf16 storage: with a prepacked weight only (as for the wide kernel)."""
    if rows_out == 0:
        return None
    lib = _hip.load()
    v = (lib.aabr_conv_wide_split_bf16 if bf16 else lib.aabr_conv_wide_split)(n_in, n_out, rows_in, rows_out, vol)
    return (v & 0xffff, v >> 16) if v else None


def _conv_dw(inp, d_out, gather, d_weight, d_bias):
    lib = _hip.load()
    if d_weight.dim() == 4 and d_weight.size(1) != 1:     # groups: one launch per group (see _conv_fwd_groups)
        G, ci, co = d_weight.size(1), d_weight.size(2), d_weight.size(3)
        for g in range(G):
            dwg = torch.zeros((d_weight.size(0), 1, ci, co), dtype=d_weight.dtype, device=d_weight.device)
            dbg = torch.zeros(co, dtype=d_bias.dtype, device=d_bias.device) if (d_bias is not None and d_bias.numel()) \
                else None
            _conv_dw(inp[:, g * ci:(g + 1) * ci].contiguous(), d_out[:, g * co:(g + 1) * co].contiguous(), gather, dwg, dbg)
            d_weight[:, g] = dwg[:, 0]
            if dbg is not None:
                d_bias[g * co:(g + 1) * co] = dbg
        return
    n_in, n_out, V_out = inp.size(1), d_out.size(1), d_out.size(0)
    assert gather.rows == V_out
    assert d_weight.is_contiguous() and d_weight.numel() == gather.vol * n_in * n_out
    pairs = gather.pairs()
    mc = gather.max_chunks(n_in, n_out)
    scratch = _hip.workspace("dw", lib.aabr_conv_dw_scratch_floats(mc, n_in, n_out), torch.float32, inp.device)
    _same_dtype(inp, d_out, "convolution backward")
    fn = lib.aabr_conv_backward_weight_bf16 if inp.dtype == torch.bfloat16 else lib.aabr_conv_backward_weight
    check(fn(ptr(inp), n_in, ptr(d_out), n_out, V_out, ptr(pairs), gather.vol, mc,
             ptr(d_weight), ptr(_opt(d_bias)), ptr(scratch), stream()))
    if trace is not None:
        trace.append(("dw", n_in, n_out, gather, inp.size(0), 0, inp.dtype))


def compile_streams(gather, rows_in, n_in, n_out, dtype, weight_grad=False):
    """Build, ahead of their first use, the block stream the forward-form launch (n_in -> n_out over `gather`) will
    read -- the same choice `_conv_fwd` makes -- and, with `weight_grad`, the offset-pair lists of the dW kernel."""
    if gather is None or gather.rows == 0:
        return
    if narrow_ok(n_in, n_out, rows_in, gather.rows, gather.vol, dtype == torch.bfloat16):   # reads the gather table itself
        if weight_grad:
            gather.pairs()
        return
    tile_rows = wide_tile_rows(n_in, n_out, rows_in, gather.rows, gather.vol, dtype == torch.bfloat16)
    sp = None if tile_rows else wide_split(n_in, n_out, rows_in, gather.rows, gather.vol, dtype == torch.bfloat16)
    if tile_rows:
        gather.blocks_wide(tile_rows)
    elif sp:
        gather.blocks_wide(sp[0])
    else:
        gather.blocks()
    if weight_grad:
        gather.pairs()


def _macs(tb, weight):
    if not count_macs:
        return 0.0
    return LazyMacs([tb.out.total_slot() + (float(weight.size(2) * weight.size(3) * weight.size(1)),)])


# SubmanifoldConvolution (pybind.cpp:134-143)
def SubmanifoldConvolution_updateOutput(spatial_size, filter_size, metadata, input_features, output_features,
                                        weight, bias, pack_t=None):
    inp = _featc(input_features, "input_features")
    tb = metadata.getSubmanifoldRuleBook(spatial_size, filter_size)
    _conv_fwd(inp, output_features, tb.V_out, tb.out, weight, bias, 0, pack_t)
    return _macs(tb, weight)


def SubmanifoldConvolution_backward(spatial_size, filter_size, metadata, input_features, d_input_features,
                                    d_output_features, weight, d_weight, d_bias, pack_t=None, need_d_input=True):
    inp = _featc(input_features, "input_features")
    d_out = _featc(d_output_features, "d_output_features")
    tb = metadata.getSubmanifoldRuleBook(spatial_size, filter_size)
    if need_d_input:  # d_in[u] = sum_k d_out[table[k'][u]] @ W[vol-1-k']^T  (flags: transpose | mirrored offset)
        _conv_fwd(d_out, d_input_features, tb.V_in, tb.out, weight, None, 1 | 2, pack_t)
    _conv_dw(inp, d_out, tb.out, d_weight, d_bias)


# Convolution (pybind.cpp:54-65)
def Convolution_updateOutput(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                             output_features, weight, bias, pack_t=None):
    inp = _featc(input_features, "input_features")
    tb = metadata.getRuleBook(input_size, output_size, filter_size, filter_stride)
    _conv_fwd(inp, output_features, tb.V_out, tb.out, weight, bias, 0, pack_t)
    return _macs(tb, weight)


def Convolution_backward(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                         d_input_features, d_output_features, weight, d_weight, d_bias, pack_t=None, need_d_input=True):
    inp = _featc(input_features, "input_features")
    d_out = _featc(d_output_features, "d_output_features")
    tb = metadata.getRuleBook(input_size, output_size, filter_size, filter_stride)
    if need_d_input:
        _conv_fwd(d_out, d_input_features, tb.V_in, tb.inn, weight, None, 1, pack_t)
    _conv_dw(inp, d_out, tb.out, d_weight, d_bias)


# Deconvolution (pybind.cpp:78-89): the rule book is looked up as (outputSize, inputSize) with
# the columns swapped (CPU/Deconvolution.cpp:15-16,34-37)
def Deconvolution_updateOutput(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                               output_features, weight, bias, pack_t=None):
    inp = _featc(input_features, "input_features")
    tb = metadata.getRuleBook(output_size, input_size, filter_size, filter_stride)
    _conv_fwd(inp, output_features, tb.V_in, tb.inn, weight, bias, 0, pack_t)
    return _macs(tb, weight)


def Deconvolution_backward(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                           d_input_features, d_output_features, weight, d_weight, d_bias, pack_t=None, need_d_input=True):
    inp = _featc(input_features, "input_features")
    d_out = _featc(d_output_features, "d_output_features")
    tb = metadata.getRuleBook(output_size, input_size, filter_size, filter_stride)
    if need_d_input:
        _conv_fwd(d_out, d_input_features, tb.V_out, tb.out, weight, None, 1, pack_t)
    _conv_dw(inp, d_out, tb.inn, d_weight, d_bias)


# ------------------------------------------------------------------------------------------------
# BatchNormalization (pybind.cpp:219-221; batchNormalization.py:120-171)
# ------------------------------------------------------------------------------------------------
def BatchNormalization_updateOutput(input_features, output_features, saveMean, saveInvStd, runningMean,
                                    runningVar, weight, bias, eps, momentum, train, leakiness):
    inp = _featc(input_features, "input_features")
    lib = _hip.load()
    output_features.resize_as_(inp)
    if inp.dim() != 2:
        return
    rows, planes = inp.shape
    for t_, nm_ in ((saveMean, "saveMean"), (saveInvStd, "saveInvStd"), (runningMean, "runningMean"),
                    (runningVar, "runningVar")):
        if t_.dtype != torch.float32 or t_.numel() < planes:
            raise TypeError("BatchNormalization: %s must hold %d float32 values, got %s[%d]"
                            % (nm_, planes, t_.dtype, t_.numel()))
    scratch = _hip.workspace("bn", lib.aabr_bn_scratch_floats(planes), torch.float32, inp.device)
    _same_dtype(inp, output_features, "BatchNormalization")
    fn = lib.aabr_bn_forward_bf16 if inp.dtype == torch.bfloat16 else lib.aabr_bn_forward
    check(fn(ptr(inp), ptr(output_features), rows, planes, ptr(saveMean), ptr(saveInvStd),
                              ptr(runningMean), ptr(runningVar), ptr(_opt(weight)), ptr(_opt(bias)), float(eps),
                              float(momentum), int(bool(train)), float(leakiness), ptr(scratch), stream()))


def BatchNormalization_backward(input_features, d_input_features, output_features, d_output_features, saveMean,
                                saveInvStd, runningMean, runningVar, weight, bias, d_weight, d_bias, leakiness):
    """NB: the reference overwrites d_output_features in place with the activation-masked
    gradient (CPU/BatchNormalization.cpp:79-82); nothing downstream reads it, so this
    implementation leaves it untouched (one HBM write pass saved)."""
    inp = _featc(input_features, "input_features")
    d_out = _featc(d_output_features, "d_output_features")
    lib = _hip.load()
    d_input_features.resize_as_(inp)
    if inp.dim() != 2:
        return
    rows, planes = inp.shape
    scratch = _hip.workspace("bn", lib.aabr_bn_scratch_floats(planes), torch.float32, inp.device)
    _same_dtype(inp, d_out, "BatchNormalization backward")
    _same_dtype(inp, output_features, "BatchNormalization backward")
    fn = lib.aabr_bn_backward_bf16 if inp.dtype == torch.bfloat16 else lib.aabr_bn_backward
    check(fn(ptr(inp), ptr(d_input_features), ptr(output_features.contiguous()), ptr(d_out),
                               rows, planes, ptr(saveMean), ptr(saveInvStd), ptr(_opt(weight)), ptr(_opt(bias)),
                               ptr(_opt(d_weight)), ptr(_opt(d_bias)), float(leakiness), ptr(scratch), stream()))


# ------------------------------------------------------------------------------------------------
# SparseToDense (pybind.cpp:124-133; SCN/CPU/SparseToDense.cpp:36-87)
# ------------------------------------------------------------------------------------------------
def _batch_size(metadata):
    """number of samples = size of the per-sample grid vector in the reference
    (`m.grids.begin()->second.size()`, SparseToDense.cpp:43); one small read-back, cached"""
    bs = getattr(metadata, "_batch_size_cache", None)
    if bs is None:
        g = metadata.grids[metadata.input_spatial]
        bs = int(g.coords[:, 3].max().item()) + 1 if g.V else 0
        metadata._batch_size_cache = bs
    return bs


def SparseToDense_updateOutput(spatial_size, metadata, input_features, output_features, nPlanes):
    inp = _f32c(input_features, "input_features")
    g = metadata.grids[_key(spatial_size)]
    sp = _key(spatial_size)
    bs = _batch_size(metadata)
    output_features.resize_(bs, int(nPlanes), *sp)
    check(_hip.load().aabr_sparse_to_dense_forward(ptr(g.coords), g.V, ptr(inp), inp.size(1) if inp.dim() == 2 else
                                                   int(nPlanes), _hip.i32x3(sp), bs, ptr(output_features),
                                                   stream()))


def SparseToDense_updateGradInput(spatial_size, metadata, input_features, d_input_features, d_output_features):
    d_out = _f32c(d_output_features, "d_output_features")
    g = metadata.grids[_key(spatial_size)]
    d_input_features.resize_as_(input_features)
    if input_features.dim() != 2:
        return
    check(_hip.load().aabr_sparse_to_dense_backward(ptr(g.coords), g.V, ptr(d_input_features),
                                                    input_features.size(1), _hip.i32x3(_key(spatial_size)),
                                                    ptr(d_out), stream()))
