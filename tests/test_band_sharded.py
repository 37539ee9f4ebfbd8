"""Band-sharded decode of one frame (BASELINE config 4 in miniature; include/jxl_amd.h "Band-sharded decode", jxl_coder_amd/shard.py).

CPU: band geometry, the halo message schedule over a world_size-2 gloo group (stub bands), the oracle against the committed row sums
of the tall fixtures.  GPU (-m gpu): a frame decoded as 2, 4 and 8 bands on ONE MI355X — the neighbour's halo handed over through
device buffers, exactly the bytes RCCL would carry — must equal the whole-frame decode bit for bit, and the reference within +-1."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, VARDCT_MAX_ABS, VARDCT_MEAN_ABS

BAND_CASES = ["vb264x4200_e7_epf3", "vb520x4400_e7"]       # 17 group rows = 3 LF-group rows; Gaborish + EPF x3 (H = 7) / x1 (H = 3)


def test_band_rows_cover_the_frame_and_prefer_lf_group_borders():
    from jxl_coder_amd.shard import band_rows, band_owner
    for ygroups in (1, 2, 7, 8, 9, 16, 17, 128, 131):
        for nbands in (1, 2, 3, 4, 8, 16):
            if nbands > ygroups:
                with pytest.raises(ValueError):
                    band_rows(ygroups, nbands)
                continue
            rows = band_rows(ygroups, nbands)
            assert rows[0][0] == 0 and rows[-1][1] == ygroups and len(rows) == nbands
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:])) and all(r1 > r0 for r0, r1 in rows)
            if (ygroups + 7) // 8 >= nbands:                        # enough LF-group rows: no LF group is split
                assert all(r0 % 8 == 0 for r0, _ in rows)
    assert band_rows(128, 8) == [(16 * b, 16 * b + 16) for b in range(8)]      # BASELINE config 4: 8 bands of 4096 rows
    for world in (1, 2, 4, 8):
        owners = [band_owner(b, 8, world) for b in range(8)]
        assert owners == sorted(owners) and set(owners) == set(range(world))


class _StubBand:
    """Stands in for DeviceBand on CPU: exports are tagged buffers, imports are recorded."""

    def __init__(self, index):
        self.index, self.got = index, {}

    def export(self, kind, side):
        return torch.full((32,), 100 * kind + 10 * self.index + side, dtype=torch.uint8)

    def recv_buffer(self, kind):
        return torch.zeros(32, dtype=torch.uint8)

    def import_(self, kind, side, buf):
        self.got[(kind, side)] = int(buf[0]) if bool((buf == buf[0]).all()) else -1


def _halo_worker(rank, world, port, nbands):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from jxl_coder_amd.shard import band_owner, exchange_halos
    bands = {b: _StubBand(b) for b in range(nbands) if band_owner(b, nbands, world) == rank}
    sent = 0
    for kind in (0, 1):
        sent += exchange_halos(bands, kind, nbands, rank, world)
    for b, band in bands.items():
        for kind in (0, 1):
            if b > 0:
                assert band.got[(kind, 0)] == 100 * kind + 10 * (b - 1) + 1, (b, band.got)      # rows above me = upper neighbour's BOTTOM edge
            else:
                assert (kind, 0) not in band.got
            if b < nbands - 1:
                assert band.got[(kind, 1)] == 100 * kind + 10 * (b + 1) + 0, (b, band.got)      # rows below me = lower neighbour's TOP edge
            else:
                assert (kind, 1) not in band.got
    t = torch.tensor([sent])
    dist.all_reduce(t)
    assert int(t) == 2 * 2 * (world - 1)          # per kind, each cross-rank border carries one message each way
    dist.destroy_process_group()


@pytest.mark.parametrize("nbands", [2, 5, 8])
def test_halo_schedule_two_ranks_gloo(nbands):
    mp.spawn(_halo_worker, args=(2, 29620 + nbands, nbands), nprocs=2, join=True)


class _SizedStubBand(_StubBand):
    """CPU tensors of the true halo sizes of an 8192-pixel-wide frame: LF halo = one cell row x 14 B (14 336 B), pixel halo = H = 3 rows
    (Gaborish + one EPF iteration) x 8192 px x 3 channels x f32 (294 912 B).  The payload is position dependent, so a truncated or
    reordered message shows."""
    SIZES = {0: 8192 // 8 * 14, 1: 3 * 8192 * 3 * 4}

    def _payload(self, kind, index, side):
        n = self.SIZES[kind]
        return ((torch.arange(n, dtype=torch.int64) * (7 + 2 * kind) + 31 * index + 5 * side) % 251).to(torch.uint8)

    def export(self, kind, side):
        return self._payload(kind, self.index, side)

    def recv_buffer(self, kind):
        return torch.zeros(self.SIZES[kind], dtype=torch.uint8)

    def import_(self, kind, side, buf):
        src = self.index - 1 if side == 0 else self.index + 1
        self.got[(kind, side)] = bool(torch.equal(buf, self._payload(kind, src, 1 - side)))


def _sized_halo_worker(rank, world, port, nbands):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from jxl_coder_amd.shard import band_owner, exchange_halos
    bands = {b: _SizedStubBand(b) for b in range(nbands) if band_owner(b, nbands, world) == rank}
    for kind in (0, 1):
        exchange_halos(bands, kind, nbands, rank, world)
    for b, band in bands.items():
        for kind in (0, 1):
            assert band.got.get((kind, 0), b == 0) is True and band.got.get((kind, 1), b == nbands - 1) is True, (b, band.got)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nbands", [(2, 8), (4, 8)])
def test_halo_messages_of_true_size_across_ranks_gloo(world, nbands):
    """BASELINE config 4's exchange step with the message sizes of an 8192-wide frame, 2 and 4 ranks (the RCCL path uses the same
    batch_isend_irecv schedule with device tensors)."""
    mp.spawn(_sized_halo_worker, args=(world, 29650 + world, nbands), nprocs=world, join=True)


def test_run_concurrently_runs_all_and_reraises():
    from jxl_coder_amd.shard import run_concurrently
    seen = []
    run_concurrently(lambda i: seen.append(i), range(5))
    assert sorted(seen) == [0, 1, 2, 3, 4]
    with pytest.raises(ZeroDivisionError):
        run_concurrently(lambda i: 1 // (i - 2), range(4))


@pytest.mark.parametrize("name", BAND_CASES)
def test_oracle_matches_reference_row_sums(oracle, golden_meta, name):
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    out, _ = oracle.decode(data, 8)
    ref = np.array(golden_meta[name]["row_sums"])
    rs = out.astype(np.int64).sum(axis=(1, 2))
    assert out.shape == tuple(golden_meta[name]["shape"])
    assert np.abs(rs - ref).max() <= VARDCT_MEAN_ABS * out.shape[1] * 4          # mean |diff| <= 0.1 per sample, row by row


# ---------------------------------------------------------------------------------------------------------------- GPU
def _bands_to_image(parts, w, h, dtype=np.uint8):
    img = np.zeros((h, w, 4), dtype)
    for y0, y1, t in parts:
        img[y0:y1] = t.cpu().numpy().view(dtype).reshape(y1 - y0, w, 4)
    return img


@pytest.mark.gpu
@pytest.mark.parametrize("name", BAND_CASES)
def test_bands_equal_whole_frame_decode(golden_meta, name):
    import jxl_coder_amd as J
    from jxl_coder_amd.shard import decode_sharded
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    dec = J.JxlDecoder(0)
    whole, info = dec.decode_one_shot(data)
    h, w = whole.shape[:2]
    ref = np.array(golden_meta[name]["row_sums"])
    assert np.abs(whole.astype(np.int64).sum(axis=(1, 2)) - ref).max() <= VARDCT_MEAN_ABS * w * 4     # whole-frame decode vs the reference
    for nbands in (2, 4, 8, 17):                        # 2: LF-group aligned; 4, 8, 17: bands share LF groups (decoded on both sides)
        parts = decode_sharded(data, nbands=nbands, rank=0, world=1, device=0)
        torch.cuda.synchronize()
        assert [p[0] for p in parts] == sorted(p[0] for p in parts) and parts[0][0] == 0 and parts[-1][1] == h
        img = _bands_to_image(parts, w, h)
        assert np.array_equal(img, whole), (nbands, np.argwhere((img != whole).any(axis=(1, 2)))[:8].ravel())
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", BAND_CASES)
def test_c_abi_sharded_local_equals_whole_frame_decode(name):
    """jxlamd_decode_sharded_local (include/jxl_amd.h): the band protocol for the bands of one process driven from C — what a C++ binding of the reference
    calls for BASELINE config 4 without a Python driver.  2, 3 and 8 bands on as many decoder contexts; band cut by jxlamd_band_rows (= shard.band_rows);
    the assembled rows equal the whole-frame decode bit for bit."""
    import ctypes as C
    import jxl_coder_amd as J
    from jxl_coder_amd.shard import band_rows
    L = J.api.lib()
    L.jxlamd_band_rows.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.jxlamd_decode_sharded_local.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    dec = J.JxlDecoder(0)
    whole, info = dec.decode_one_shot(data)
    h, w = whole.shape[:2]
    ygroups = (h + 255) // 256
    decs = [J.JxlDecoder(0) for _ in range(8)]
    for nbands in (2, 3, 8):
        rows = (C.c_int * (2 * nbands))()
        assert L.jxlamd_band_rows(ygroups, nbands, rows) == 0
        assert [(rows[2 * b], rows[2 * b + 1]) for b in range(nbands)] == band_rows(ygroups, nbands)
        outs = [torch.zeros((min(rows[2 * b + 1] * 256, h) - rows[2 * b] * 256) * w * 4, dtype=torch.uint8, device="cuda:0") for b in range(nbands)]
        torch.cuda.synchronize()
        hs = (C.c_void_p * nbands)(*[d._h for d in decs[:nbands]])
        ps = (C.c_void_p * nbands)(*[o.data_ptr() for o in outs])
        caps = (C.c_size_t * nbands)(*[o.numel() for o in outs])
        rc = L.jxlamd_decode_sharded_local(hs, nbands, data, len(data), J.api.JXLAMD_OUT_DEVICE, ps, caps, None)
        assert rc == 0, L.jxlamd_last_error(None)
        torch.cuda.synchronize()
        img = np.concatenate([o.cpu().numpy().reshape(-1, w, 4) for o in outs])
        assert np.array_equal(img, whole), nbands
    for d in decs + [dec]:
        d.close()


@pytest.mark.gpu
def test_sharded_large_frame_against_the_reference():
    """A 2048x4352 frame (17 group rows, 8 x 17 groups, 3 LF groups) from the reference's encoder, decoded as 2 / 4 / 8 bands."""
    sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import jxl_ref
    if not jxl_ref.available():
        pytest.skip("oracle/_ref (the reference's libjxl) did not travel to this box")
    import synth
    import jxl_coder_amd as J
    from jxl_coder_amd.shard import decode_sharded
    w, h = 2048, 4352
    data = jxl_ref.encode(synth.photo_like(w, h, seed=31), effort=7, distance=1.0)
    ref = jxl_ref.decode(data, threads=0)[0]
    dec = J.JxlDecoder(0)
    whole, _ = dec.decode_one_shot(data)
    d = np.abs(whole.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= VARDCT_MAX_ABS and d.mean() <= VARDCT_MEAN_ABS
    for nbands in (2, 4, 8):
        img = _bands_to_image(decode_sharded(data, nbands=nbands, rank=0, world=1, device=0), w, h)
        assert np.array_equal(img, whole), nbands
    dec.close()


@pytest.mark.gpu
def test_band_protocol_errors_are_loud():
    import jxl_coder_amd as J
    data = open(os.path.join(GOLDEN, "vb520x4400_e7.jxl"), "rb").read()
    dec = J.JxlDecoder(0)
    out = torch.empty(520 * 4400 * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        dec.band_reconstruct()                                            # no band in progress
    with pytest.raises(ValueError):
        dec.band_begin(data, 5, 40, out.data_ptr(), out.numel())          # rows outside the frame (18 group rows)
    with pytest.raises(ValueError):
        dec.band_begin(data, 0, 8, out.data_ptr(), 1000)                  # output too small for the band
    small = open(os.path.join(GOLDEN, "l512_e7.jxl"), "rb").read()
    with pytest.raises(J.UnsupportedJXLFeature):
        dec.band_begin(small, 0, 1, out.data_ptr(), out.numel())          # Modular frame: not band-decodable
    dec.band_begin(data, 0, 8, out.data_ptr(), out.numel())
    with pytest.raises(ValueError):
        dec.band_finish()                                                 # finish before reconstruct
    whole, _ = dec.decode_one_shot(data)                                  # the context still decodes whole frames afterwards
    assert whole.shape == (4400, 520, 4)
    dec.close()


@pytest.mark.gpu
def test_band_decode_across_the_kernel_switch_at_16384_squared():
    """BASELINE config 4 at a quarter of its size (16384 x 16384 = 268 MP, 4096 groups, generated and encoded on the box by the reference's
    encoder): decoded as ONE band of 4096 groups — at and above that count a band takes the lane-per-group PassGroup kernels (k_pass_prep +
    k_pass_flat) — and as three bands of <= 1408 groups (wave-per-group kernel, LF groups split between bands): both within max 1 / mean 0.05
    of the reference's libjxl run live, and equal to each other bit for bit (tools/gpu/c4_full.py; the full 32768 x 32768 run of the same
    tool is kept as profiles/r03_c4_32768.txt)."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jxl_ref
    if not jxl_ref.available():
        pytest.skip("the reference encoder (oracle/_ref) did not travel to this box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu", "c4_full.py"), "16384", "16384", "1"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "1 bands == 3 bands" in r.stdout and "bit for bit" in r.stdout and r.stdout.rstrip().endswith("True"), r.stdout[-800:]


@pytest.mark.gpu
def test_config4_at_its_true_size_last_band():
    """BASELINE configs[3] at its true size — one 32768 x 32768 VarDCT frame (16 384 groups, 4 GiB of RGBA8), generated and encoded on the box by
    the reference's encoder: the LAST of the eight bands (group rows 112 - 128, byte offset 3.76 GB of the frame's output: beyond 2^31 and 2^32 / 1.14)
    decoded next to its upper neighbour, once as one band and once as three (borders inside LF groups): within max 1 / mean 0.05 of the reference's
    libjxl run live on the same file, and bit-identical between the two partitions (tools/gpu/c4_full.py ... tail)."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jxl_ref
    if not jxl_ref.available():
        pytest.skip("the reference encoder (oracle/_ref) did not travel to this box")
    free_gb = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) / 1e6
    if free_gb < 60:
        pytest.skip("needs ~52 GB of host RAM for the 1 GP encode")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu", "c4_full.py"), "32768", "32768", "8", "tail"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "bit for bit" in r.stdout and r.stdout.rstrip().endswith("True") and "frame output byte offset 3758096384" in r.stdout, r.stdout[-800:]
