"""Multi-GPU sharding of the decode path, one process per GPU.

* Batches (BASELINE configs 3/5): independent frames shard round-robin over ranks; NO data-path collective.
  torch.distributed is used only for the barrier / max-over-ranks timing.
* One huge frame (BASELINE config 4): contiguous bands of 256x256 group rows per rank (include/jxl_amd.h "Band-sharded decode").
  PassGroups decode and reconstruct independently; the only exchange is the halo at band borders — one cell row of LF data after the
  LF stage and H = 1 (Gaborish) + 3/2/1 (EPF iterations) pre-filter pixel rows after reconstruction — moved with point-to-point
  send/recv in one group (torch.distributed.batch_isend_irecv = ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd over xGMI);
  bands that live on the same GPU hand their halo buffers over directly.

The reference decodes a frame in one process (libjxl under interop/JxlDecoding.cpp:75); this module is the "decode_sharded" extension
of SURVEY.md §8b.
"""
HALO_LF, HALO_PIXELS = 0, 1


def shard_indices(n_items: int, rank: int, world: int):
    return list(range(rank, n_items, world))


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---------------------------------------------------------------------------------------------------------------- band geometry
def band_rows(ygroups: int, nbands: int):
    """[(group_row0, group_row1)] of `nbands` contiguous bands covering [0, ygroups).  Borders fall on multiples of 8 group rows
    (= 2048-pixel LF groups stay whole, no LF stream is decoded twice) whenever there are at least `nbands` LF-group rows."""
    if nbands < 1 or nbands > ygroups:
        raise ValueError(f"cannot cut {ygroups} group rows into {nbands} bands")
    lf_rows = (ygroups + 7) // 8
    unit, units = (8, lf_rows) if lf_rows >= nbands else (1, ygroups)
    cuts = [min(ygroups, (units * b // nbands) * unit) for b in range(nbands + 1)]
    cuts[-1] = ygroups
    return [(cuts[b], cuts[b + 1]) for b in range(nbands)]


def band_owner(band: int, nbands: int, world: int) -> int:
    """Contiguous blocks of bands per rank: neighbours mostly share a GPU, so at most world-1 borders cross xGMI."""
    return band * world // nbands


# ---------------------------------------------------------------------------------------------------------------- halo schedule
def exchange_halos(bands, kind: int, nbands: int, rank: int, world: int, group=None):
    """One halo exchange step for the bands this rank holds.

    `bands`: {band index: object with export(kind, side) -> 1-D uint8 tensor, import_(kind, side, tensor), recv_buffer(kind) -> tensor}.
    Border between band b and b+1: b's bottom edge (side 1) becomes the rows above b+1 (its side 0) and b+1's top edge (side 0)
    becomes the rows below b (its side 1).  Same-rank borders are handed over directly; the others go into ONE batch of
    isend/irecv ops (a single ncclGroup), ordered identically on both ends.  Returns the number of messages this rank sent."""
    import torch.distributed as dist
    ops, pending, keep, sent = [], [], [], 0
    for b in range(nbands - 1):
        up, dn = band_owner(b, nbands, world), band_owner(b + 1, nbands, world)
        if up != rank and dn != rank:
            continue
        if up == dn:
            bands[b + 1].import_(kind, 0, bands[b].export(kind, 1))
            bands[b].import_(kind, 1, bands[b + 1].export(kind, 0))
            continue
        if up == rank:                                   # I hold the upper band: send its bottom edge down, receive the lower band's top edge
            out, inn = bands[b].export(kind, 1), bands[b].recv_buffer(kind)
            ops += [dist.P2POp(dist.isend, out, dn, group), dist.P2POp(dist.irecv, inn, dn, group)]
            pending.append((b, 1, inn))
        else:                                            # I hold the lower band
            out, inn = bands[b + 1].export(kind, 0), bands[b + 1].recv_buffer(kind)
            ops += [dist.P2POp(dist.irecv, inn, up, group), dist.P2POp(dist.isend, out, up, group)]
            pending.append((b + 1, 0, inn))
        keep.append(out); sent += 1
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if keep[0].is_cuda:
            import torch
            torch.cuda.synchronize()                     # the imports run on the decoder contexts' own HIP streams
    for b, side, buf in pending:
        bands[b].import_(kind, side, buf)
    return sent


class DeviceBand:
    """One band of a frame on this rank's GPU: a decoder context + its output rows + halo buffers (all HBM-resident)."""

    def __init__(self, decoder, data: bytes, rows, width: int, height: int, bytes_per_pixel: int, device, out=None, begin=True, shared_gpu=False):
        import torch
        self.dec, self.rows, self.data = decoder, rows, data
        self.py0, self.py1 = rows[0] * 256, min(rows[1] * 256, height)
        n = (self.py1 - self.py0) * width * bytes_per_pixel
        self.out = out if out is not None else torch.empty(n, dtype=torch.uint8, device=device)      # `out`: the caller's (reused) rows
        assert self.out.numel() >= n
        self._bufs = {}
        self.device = device
        self.info = None
        self.shared_gpu = shared_gpu        # other bands of the frame run on this GPU at the same time (JXLAMD_BAND_SHARED_GPU)
        if begin:
            self.begin()

    def begin(self):
        """parse + upload + LF stage of the band (returns when done; bands of one GPU call this from one thread each)"""
        self.info = self.dec.band_begin(self.data, self.rows[0], self.rows[1], self.out.data_ptr(), self.out.numel(), shared_gpu=self.shared_gpu)

    def _buffer(self, kind, slot):
        import torch
        key = (kind, slot)
        if key not in self._bufs:
            self._bufs[key] = torch.empty(max(self.dec.band_halo_bytes(kind), 16), dtype=torch.uint8, device=self.device)
        return self._bufs[key]

    def export(self, kind, side):
        buf = self._buffer(kind, ("out", side))
        self.dec.band_export(kind, side, buf.data_ptr(), buf.numel())
        return buf

    def recv_buffer(self, kind):
        self._n = getattr(self, "_n", 0) + 1
        return self._buffer(kind, ("in", self._n))

    def import_(self, kind, side, buf):
        self.dec.band_import(kind, side, buf.data_ptr(), buf.numel())


_contexts = {}


def run_concurrently(fn, items):
    """fn(item) for every item, one host thread each (the C-ABI calls release the GIL): the bands a GPU holds go through each protocol
    phase side by side on their own decoder contexts / HIP streams, like the flights of a batch — their LF stages are ~150 ms of latency
    each and leave the chip almost empty when run one after another."""
    items = list(items)
    if len(items) <= 1:
        for it in items:
            fn(it)
        return
    import threading
    errors = []

    def wrap(it):
        try:
            fn(it)
        except BaseException as e:  # noqa: BLE001 — re-raised below
            errors.append(e)
    th = [threading.Thread(target=wrap, args=(it,)) for it in items]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]


def decode_sharded(data: bytes, nbands=None, rank: int = 0, world: int = 1, device: int = 0, group=None, allowed_floats=True, outs=None):
    """Decode ONE frame as `nbands` bands (default: one per rank).  Every rank calls this with the same bytes; returns
    [(pixel_row0, pixel_row1, uint8 CUDA tensor of those rows, tight RGBA8/RGBA16)] for the bands this rank decoded.  The pixels equal
    the same rows of a whole-frame decode bit for bit.  outs: optional list of preallocated uint8 CUDA tensors, one per band of this rank."""
    from . import api
    info = api.Info()
    rc = api.lib().jxlamd_basic_info(data, len(data), api.C.byref(info))
    if rc:
        raise api.InvalidJXLException(api.lib().jxlamd_last_error(None).decode())
    w, h = info.xsize, info.ysize
    bpp = 8 if (info.out_bits == 16 and allowed_floats) else 4
    ygroups = (h + 255) // 256
    nbands = world if nbands is None else nbands
    rows = band_rows(ygroups, nbands)
    mine = [b for b in range(nbands) if band_owner(b, nbands, world) == rank]
    dev = f"cuda:{device}"
    bands = {}
    for k, b in enumerate(mine):
        key = (device, k)
        if key not in _contexts:
            _contexts[key] = api.JxlDecoder(device)
        bands[b] = DeviceBand(_contexts[key], data, rows[b], w, h, bpp, dev, out=outs[k] if outs else None, begin=False, shared_gpu=len(mine) > 1)
    # every phase of the protocol: the bands this GPU holds side by side (one decoder context + host thread each), then the halo step.
    # (The pixel halo is produced by the reconstruction of the band's border groups, i.e. by the phase it follows: there is no interior
    # work left to overlap it with — PassGroup decode and inverse DCT of ALL groups precede the filters; the messages are 1.2 MB.)
    import os, time
    trace = os.environ.get("JXLAMD_TRACE_BANDS")
    t0 = time.perf_counter()
    run_concurrently(lambda b: bands[b].begin(), mine)                              # parse + LF stage
    t1 = time.perf_counter()
    exchange_halos(bands, HALO_LF, nbands, rank, world, group)
    t2 = time.perf_counter()
    run_concurrently(lambda b: bands[b].dec.band_reconstruct(), mine)
    t3 = time.perf_counter()
    exchange_halos(bands, HALO_PIXELS, nbands, rank, world, group)
    t4 = time.perf_counter()
    run_concurrently(lambda b: bands[b].dec.band_finish(), mine)
    if trace:
        t5 = time.perf_counter()
        print("[bands rank %d] begin %.1f | LF halos %.1f | reconstruct %.1f | pixel halos %.1f | finish %.1f ms" % (
            rank, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3), flush=True)
    return [(bands[b].py0, bands[b].py1, bands[b].out) for b in mine]
