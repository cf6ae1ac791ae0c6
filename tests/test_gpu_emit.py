"""Parity tests for K3 (RBSP -> Annex-B) and the device generator of the
synthetic workload, through the C ABI, against the oracle."""
import numpy as np
import pytest

from tests._orc import NAL_ENTRY

pytestmark = pytest.mark.gpu
ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


@pytest.fixture(scope="module")
def ctx():
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    yield c
    c.close()


@pytest.fixture(params=[-1, 0, 1, 2], ids=["auto", "single-pass", "three-step", "arena-tiles"])
def path_ctx(ctx, request):
    """every way hbs_emit_annexb can run; auto sends a handful of small NALs through the one-launch kernel"""
    ctx.set_emit_path(request.param)
    yield ctx
    ctx.set_emit_path(-1)


def fake_index(lens, gaps):
    idx = np.zeros(len(lens), dtype=NAL_ENTRY)
    off = pos = 0
    for k, (n, g) in enumerate(zip(lens, gaps)):
        idx["start"][k] = pos + g
        idx["end"][k] = pos + g + n
        idx["rbsp_off"][k] = off
        idx["rbsp_len"][k] = n
        pos += g + n
        off += n
    return idx


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_emit_random_rbsp(path_ctx, orc):
    ctx = path_ctx
    rng = np.random.RandomState(31)
    for _ in range(60):
        nn = rng.randint(1, 40)
        lens = [int(rng.randint(1, 3000)) for _ in range(nn)]
        gaps = [int(rng.randint(3, 7)) for _ in range(nn)]
        arena = ALPHA[rng.randint(0, len(ALPHA), size=sum(lens))].copy()
        if rng.rand() < 0.5:
            arena[rng.rand(len(arena)) < 0.6] = 0x55
        idx = fake_index(lens, gaps)
        got, _ = ctx.emit_annexb(dev(arena), idx)
        assert np.array_equal(got, orc.emit_annexb(arena, idx))


def test_emit_row_and_batch_edges(path_ctx, orc):
    ctx = path_ctx
    """NAL lengths around the kernel's row (1 KiB) and register batch (16 KiB) sizes, mostly-plain bytes with a few
    zero runs (the register copy path), the arena ending exactly at the last NAL's last byte"""
    rng = np.random.RandomState(32)
    edge = [1, 15, 16, 17, 1023, 1024, 1025, 1039, 1040, 16383, 16384, 16385, 16400, 32768, 32769, 40000, 70001]
    for rep in range(6):
        lens = [int(edge[i]) for i in rng.permutation(len(edge))[:rng.randint(3, len(edge))]]
        gaps = [int(rng.randint(3, 6)) for _ in lens]
        arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
        for p in rng.randint(0, len(arena), size=len(arena) // 700 + 2):
            arena[p:p + rng.randint(2, 5)] = 0
            if rng.rand() < 0.5 and p + 5 < len(arena):
                arena[p + 4] = rng.randint(0, 5)
        idx = fake_index(lens, gaps)
        got, _ = ctx.emit_annexb(dev(arena), idx)
        want = orc.emit_annexb(arena, idx)
        assert np.array_equal(got, want), (rep, lens)


def test_emit_long_nals(ctx, orc):
    """coded pictures of 50 KiB .. 3 MiB between small NALs: a long NAL is split into 12 KiB segments that different
    wavefronts and workgroups write; zero runs straddle the segment borders"""
    rng = np.random.RandomState(33)
    for rep in range(3):
        lens = [int(x) for x in rng.permutation([7, 300, 12288, 12289, 24576, 24577, 50_000, 123_457, 500_001, 3_000_000 if rep == 0 else 200_000, 40, 9000])]
        gaps = [int(rng.randint(3, 6)) for _ in lens]
        arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
        pos = 0
        for n in lens:                                       # zeros across every 12 KiB border of every NAL
            for b in range(12288, n, 12288):
                arena[pos + b - rng.randint(1, 4): pos + b + rng.randint(0, 3)] = 0
            pos += n
        for p in rng.randint(0, len(arena), size=len(arena) // 5000 + 2):
            arena[p:p + rng.randint(2, 6)] = 0
        idx = fake_index(lens, gaps)
        got, got_idx = ctx.emit_annexb(dev(arena), idx)
        want = orc.emit_annexb(arena, idx)
        assert np.array_equal(got, want), (rep, lens)
        pos = 0
        for k, (n, g) in enumerate(zip(lens, gaps)):         # the output index: where each NAL went
            start = pos + g
            assert int(got_idx["start"][k]) == start and bytes(want[start - 3:start]) == b"\x00\x00\x01"
            pos = int(got_idx["end"][k])
        assert pos == len(want)


def test_emit_arena_tile_edges(ctx, orc):
    """the arena-tile kernel (path 2) on what its geometry makes special -- NALs that begin on, one byte before and one byte behind a
    192 KiB tile border and a 16-byte chunk border, empty NALs alone, in a row, at the very start and at the very end, an arena that
    ends on a tile border / a chunk border / in the middle of a chunk, zeros on both sides of a NAL border (a start restarts the
    count: no 03 across it), gaps from 3 to 300 bytes -- and on indexes it must hand to the kernel by NALs (a first byte that is not
    16-byte aligned, a gap of 1 MiB, more than 256 NALs in one tile).  Bytes against the oracle, output index against path 0."""
    T = 192 * 1024
    rng = np.random.RandomState(35)
    cases = []
    # lengths that put NAL starts around tile and chunk borders; arena ends on a tile border, a chunk border, mid-chunk
    for tail in (0, 16, 7):
        lens = [0, 5, 11, 15, 1, T - 32 - 1, 1, 1, 0, 0, 0, 30, T - 30 - 16, 16, 17, 0, 2 * T - 33 + tail, 0, 0]
        cases.append((lens, [int(rng.randint(3, 12)) for _ in lens], 0))
    cases.append(([T, T, T], [3, 4, 300], 0))
    cases.append(([100, 200, 300000], [3, 3, 3], 8))                       # first byte at arena offset 8: not eligible
    cases.append(([5000, 5000, 400000], [3, 1 << 20, 4], 0))               # a gap of 1 MiB: not eligible
    cases.append(([64] * 700 + [300000], [3] * 701, 0))                    # 700 NALs in the first tile: not eligible
    for lens, gaps, lead in cases:
        arena = rng.randint(1, 256, size=lead + sum(lens)).astype(np.uint8)
        pos = lead
        for n in lens:                                                     # zeros on both sides of every border, and a few inside
            arena[max(pos - 3, 0): pos + 3] = 0
            pos += n
        for q in rng.randint(0, max(len(arena) - 8, 1), size=len(arena) // 3000 + 2):
            arena[q:q + rng.randint(2, 6)] = 0
            arena[q + 5] = rng.randint(0, 5)
        idx = fake_index(lens, gaps)
        idx["rbsp_off"] += lead
        want = orc.emit_annexb(arena, idx)
        ctx.set_emit_path(0)
        by_nals, idx_nals = ctx.emit_annexb(dev(arena), idx)
        ctx.set_emit_path(2)
        by_tiles, idx_tiles = ctx.emit_annexb(dev(arena), idx)
        ctx.set_emit_path(-1)
        assert np.array_equal(by_nals, want), (lens[:6], "by NALs")
        assert np.array_equal(by_tiles, want), (lens[:6], "by arena tiles")
        assert np.array_equal(idx_tiles, idx_nals), (lens[:6], "output index")


@pytest.mark.parametrize("shape", ["zeros-10-percent", "nals-600", "insertions-every-500"])
def test_emit_arena_tiles_with_several_batches_of_elements(ctx, orc, shape):
    """65 to 1024 elements per 192 KiB tile of the arena (round 3: exact flags, wavefront 1 takes every other batch, batch sums in
    LDS): zero-heavy payload, NALs of ~600 bytes, a 03 to insert every ~500 bytes.  Arena tiles (pinned), the automatic choice and
    the kernel by NALs against the oracle; the output indexes against each other."""
    rng = np.random.RandomState({"zeros-10-percent": 41, "nals-600": 42, "insertions-every-500": 43}[shape])
    total = 2 * (1 << 20) + 777
    if shape == "nals-600":
        lens = []
        while sum(lens) < total:
            lens.append(int(rng.randint(400, 800)))
    else:
        lens = [int(x) for x in rng.randint(3000, 40000, size=total // 21500)]
    arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
    if shape == "zeros-10-percent":
        arena[rng.random_sample(len(arena)) < 0.10] = 0
    elif shape == "insertions-every-500":
        for q in rng.randint(0, len(arena) - 8, size=len(arena) // 500):
            arena[q:q + 3] = (0, 0, rng.randint(0, 4))
    idx = fake_index(lens, [int(rng.randint(3, 5)) for _ in lens])
    want = orc.emit_annexb(arena, idx)
    outs = {}
    for path in (0, 2, -1):
        ctx.set_emit_path(path)
        outs[path] = ctx.emit_annexb(dev(arena), idx)
    ctx.set_emit_path(-1)
    for path, (got, got_idx) in outs.items():
        assert np.array_equal(got, want), (shape, path)
        assert np.array_equal(got_idx, outs[0][1]), (shape, path, "output index")


def test_emit_arena_tiles_walk_dense_tiles_by_rows(ctx, orc):
    """an arena that is sparse but for stretches the density probe does not see -- 00 00 03 padding, pure zeros (what
    cabac_zero_words leave in an RBSP), zeros with a NAL beginning inside them, runs that cross wavefront and tile
    boundaries with odd and even lengths: the tiles that hold them have thousands of elements and are walked by rows
    (k3_dense_tile; until round 4 the tile kernel gave the whole call up).  Bytes against the oracle's rbsp_to_nal, the
    output index against the kernel by NALs, and the tile kernel must have done the whole call."""
    T = 192 * 1024
    W = 48 * 1024
    rng = np.random.RandomState(36)
    cases = [("pad", 3 * T + 1000, 150_000), ("pad", 100, 30_000), ("pad", 9 * T - 40_000, 80_000),
             ("zero", 2 * T + 5000, 200_000), ("zero", 4 * T - 7, 50_001), ("zero", 5 * T + W - 3, 40_006), ("zero", 16, T + 16),
             ("zero", 7 * T + 1, 2 * T + 3), ("mix", 6 * T + 123, 120_000)]
    for kind, dense_at, dense_len in cases:
        lens = [int(x) for x in rng.randint(2000, 60000, size=70)]
        arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
        if kind == "pad":
            arena[dense_at: dense_at + dense_len] = np.tile(np.array([0, 0, 3], dtype=np.uint8), dense_len // 3 + 1)[:dense_len]
        elif kind == "zero":
            arena[dense_at: dense_at + dense_len] = 0
        else:                                            # zeros and small values, runs of every length
            arena[dense_at: dense_at + dense_len] = ALPHA[rng.randint(0, len(ALPHA), size=dense_len)]
        idx = fake_index(lens, [int(rng.randint(3, 6)) for _ in lens])
        want = orc.emit_annexb(arena, idx)
        ctx.set_emit_path(0)
        _, idx_nals = ctx.emit_annexb(dev(arena), idx)
        ctx.set_emit_path(2)
        by_tiles, idx_tiles = ctx.emit_annexb(dev(arena), idx)
        stayed = ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h)
        ctx.set_emit_path(-1)
        assert np.array_equal(by_tiles, want), (kind, dense_at, dense_len, int(np.flatnonzero(by_tiles[:len(want)] != want[:len(by_tiles)])[0]) if len(by_tiles) == len(want) else (len(by_tiles), len(want)))
        assert np.array_equal(idx_tiles, idx_nals), (kind, dense_at)
        assert stayed == 1, (kind, dense_at)
    # NALs that begin inside a stretch of zeros, at chunk edges and inside chunks
    lens = [50_000, 100_000, 17, 16, 1, 40_000, 300_000, 33, 250_000]
    arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
    arena[30_000: 30_000 + 400_000] = 0
    idx = fake_index(lens, [3, 4, 3, 5, 3, 4, 3, 3, 4])
    want = orc.emit_annexb(arena, idx)
    ctx.set_emit_path(2)
    got, got_idx = ctx.emit_annexb(dev(arena), idx)
    stayed = ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h)
    ctx.set_emit_path(0)
    _, idx_nals = ctx.emit_annexb(dev(arena), idx)
    ctx.set_emit_path(-1)
    assert np.array_equal(got, want) and np.array_equal(got_idx, idx_nals) and stayed == 1
    # one NAL with 70 tiles of nothing but zeros in it (runs of odd and of even length): a tile learns the count it is entered
    # with from the words of the tiles in front (dz_entry_count), more than one window of 64 of them back
    for z0, z1 in ((5_000, 13_900_000), (5_001, 13_900_000), (16 * 1024, 70 * T + 16 * 1024)):
        lens = [10_000, 14_000_000, 5_000]
        arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
        arena[10_000 + z0: 10_000 + z1] = 0
        idx = fake_index(lens, [4, 3, 4])
        want = orc.emit_annexb(arena, idx)
        ctx.set_emit_path(2)
        got, got_idx = ctx.emit_annexb(dev(arena), idx)
        stayed = ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h)
        ctx.set_emit_path(-1)
        assert np.array_equal(got, want), (z0, z1)
        assert stayed == 1


def test_emit_arena_tiles_count_dense_tiles_ahead(ctx, orc):
    """round 4: k3t_sample looks at 64 bytes in every 64 KiB of a tile (sectors 1, 5, 9 of twelve), then at 64 in every 16 KiB of
    a tile that shows something, and lists the tile when 8 of those 48 chunks end a pattern 00 00 {<= 3}; the listed tiles' first
    halves are counted ahead of the main pass.  Tiles that are listed AND dense (a stretch of padding), listed and NOT dense
    (patterns exactly where the sample looks, nowhere else: the count-ahead pass drops them), dense and NOT listed (a stretch of
    20 KiB between the sampled sectors' rows is still counted in place) -- bytes against the oracle, the tile kernel did the call."""
    T = 192 * 1024
    rng = np.random.RandomState(97)

    def sample_at(tile, sec, sub):                       # hbs_emit.hip: k3t_sample
        return tile * T + 16384 * sec + 1024 * ((5 * sec + tile) & 15) + 64 * ((7 * sec + (tile >> 2)) & 15) + 16 * sub

    lens = [int(x) for x in rng.randint(20000, 90000, size=40)]
    arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
    ntiles = len(arena) // T
    assert ntiles >= 9
    for tile in (2, 5):                                  # listed, not dense: 48 chunks with a pattern, the rest of the tile clean
        for sec in range(12):
            for sub in range(4):
                x = sample_at(tile, sec, sub)
                arena[x + 5: x + 8] = (0, 0, 1)
    arena[3 * T + 1000: 3 * T + 1000 + 150_000] = np.tile(np.array([0, 0, 3], dtype=np.uint8), 50_000)      # listed and dense
    arena[7 * T + 70_000: 7 * T + 90_000] = 0                                                               # dense (a full row of zeros), seen or not
    idx = fake_index(lens, [3 + (k & 1) for k in range(len(lens))])
    want = orc.emit_annexb(arena, idx)
    ctx.set_count_ahead(2)                               # (round 6: by default only arenas from 3 GiB up are sampled, like the scan's streams)
    ctx.set_emit_path(2)
    got, got_idx = ctx.emit_annexb(dev(arena), idx)
    stayed = ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h)
    ctx.set_emit_path(0)
    _, idx_nals = ctx.emit_annexb(dev(arena), idx)
    ctx.set_emit_path(-1)
    assert np.array_equal(got, want) and np.array_equal(got_idx, idx_nals) and stayed == 1
    # the same arena twice more on the same context: the table's stamps are this call's only
    ctx.set_emit_path(2)
    arena2 = arena.copy()
    arena2[3 * T + 1000: 3 * T + 1000 + 150_000] = rng.randint(1, 256, size=150_000)     # the stretch is gone: a stale entry would be wrong
    want2 = orc.emit_annexb(arena2, idx)
    got2, _ = ctx.emit_annexb(dev(arena2), idx)
    got3, _ = ctx.emit_annexb(dev(arena), idx)
    ctx.set_count_ahead(1)
    got4, _ = ctx.emit_annexb(dev(arena), idx)           # ... and counted in place (the default at this size)
    ctx.set_emit_path(-1)
    assert np.array_equal(got2, want2) and np.array_equal(got3, want) and np.array_equal(got4, want)


def test_emit_arena_tiles_refuse_an_index_outside_the_arena(ctx, orc):
    """round 2's advice: the arena-tile kernel reads its rows unpredicated, so an index that points past the caller's RBSP
    buffer (built by hand, or corrupt) must not reach it even when the path is pinned: k3t_check compares the index with
    rbsp_bytes (every entry: the call ends with HBS_E_ARG before any kernel follows the index into the arena, whichever emit
    path is pinned); arenas shorter than one 16-byte chunk go to the kernel by NALs and are right"""
    import hevcbitstream_amd as hbs
    rng = np.random.RandomState(37)
    lens = [int(x) for x in rng.randint(2000, 60000, size=40)]
    arena = rng.randint(1, 256, size=sum(lens)).astype(np.uint8)
    good = fake_index(lens, [4] * len(lens))
    ctx.set_emit_path(2)
    try:
        want = orc.emit_annexb(arena, good)
        got, _ = ctx.emit_annexb(dev(arena), good)
        assert np.array_equal(got, want)
        bads = []
        for shift in (1 << 20, 1 << 33, (1 << 64) - (1 << 20)):
            bad = good.copy()
            bad["rbsp_off"] = bad["rbsp_off"] + np.uint64(shift)          # contiguous, aligned -- and outside the buffer
            bads.append(bad)
        bad = good.copy()
        bad["rbsp_len"][-1] += 4096                                       # the last NAL runs past the end
        bads.append(bad)
        bad = good.copy()
        bad["rbsp_off"][7] = np.uint64(1 << 40)                           # one entry in the middle
        bads.append(bad)
        for path in (2, 0, 1, -1):
            ctx.set_emit_path(path)
            for bad in bads:
                with pytest.raises(hbs.HbsError):                         # HBS_E_ARG, and nothing was read through the index
                    ctx.emit_annexb(dev(arena), bad, out_cap=2 * len(arena))
            few = fake_index([100, 200], [4, 3])
            few["rbsp_off"][1] = np.uint64(1 << 33)
            with pytest.raises(hbs.HbsError):                             # the one-launch path (auto) checks too
                ctx.emit_annexb(dev(arena[:300]), few)
            got, _ = ctx.emit_annexb(dev(arena), good)                    # the context is fine afterwards
            assert np.array_equal(got, want)
        ctx.set_emit_path(2)
        for total in (1, 5, 15, 16, 17):
            small = ALPHA[rng.randint(0, len(ALPHA), size=total)].copy()
            idx = fake_index([total], [3])
            got, _ = ctx.emit_annexb(dev(small), idx)
            assert np.array_equal(got, orc.emit_annexb(small, idx)), total
    finally:
        ctx.set_emit_path(-1)


def test_emit_zero_runs(path_ctx, orc):
    ctx = path_ctx
    for z in (2, 3, 4, 5, 255, 256, 257, 513, 70000):
        for tail in ([], [1], [4], [0, 0, 1]):
            arena = np.array([7] * 3 + [0] * z + tail, dtype=np.uint8)
            idx = fake_index([len(arena)], [4])
            got, _ = ctx.emit_annexb(dev(arena), idx)
            assert np.array_equal(got, orc.emit_annexb(arena, idx)), (z, tail)


def test_emit_few_small_nals(ctx, orc):
    """the one-launch path and its limits (256 NALs, 32 KiB): counts and sizes on both sides of them, empty NALs,
    the output index, a capacity that is too small"""
    import hevcbitstream_amd as hbs
    rng = np.random.RandomState(34)
    for nn, total in ((1, 1), (1, 32768), (1, 32769), (4, 32768), (5, 20000), (255, 9000), (256, 32768), (257, 9000), (64, 0), (3, 2)):
        cuts = np.sort(rng.randint(0, total + 1, size=nn - 1)) if nn > 1 else np.zeros(0, dtype=np.int64)
        lens = [int(x) for x in np.diff(np.concatenate(([0], cuts, [total])))]
        gaps = [int(rng.randint(3, 6)) for _ in lens]
        arena = ALPHA[rng.randint(0, len(ALPHA), size=total)].copy()
        idx = fake_index(lens, gaps)
        got, got_idx = ctx.emit_annexb(dev(arena) if total else dev(np.zeros(1, dtype=np.uint8))[:0], idx)
        want = orc.emit_annexb(arena, idx)
        assert np.array_equal(got, want), (nn, total)
        pos = 0
        for k, (n, g) in enumerate(zip(lens, gaps)):
            assert int(got_idx["start"][k]) == pos + g and int(got_idx["rbsp_len"][k]) == n
            pos = int(got_idx["end"][k])
        assert pos == len(want)
    arena = np.zeros(300, dtype=np.uint8)
    idx = fake_index([100, 200], [4, 3])
    with pytest.raises(hbs.HbsError):
        ctx.emit_annexb(dev(arena), idx, out_cap=310)      # 7 + 300 + 149 inserted bytes do not fit


def test_emit_arenas_of_tiny_nals(ctx, orc):
    """the automatic path on arenas whose NALs are too small for the arena tiles (mean below 448 bytes: a lane per NAL,
    k3_count_tiny / k3_emit_tiny): zero-heavy and plain bytes, empty NALs, NALs around the 16-byte load and store steps, gaps
    of 3-6 bytes, the output index; against the oracle's rbsp_to_nal per NAL.  Also a capacity that is too small."""
    import hevcbitstream_amd as hbs
    rng = np.random.RandomState(77)
    for mean, nn, zeros in ((40, 3000, True), (130, 2000, False), (300, 1500, True), (17, 5000, True), (1, 400, True)):
        lens = [int(x) for x in rng.randint(0, 2 * mean + 1, size=nn)]
        lens[0] = 15; lens[1] = 16; lens[2] = 17; lens[3] = 0; lens[4] = 33
        gaps = [int(rng.randint(3, 7)) for _ in lens]
        total = sum(lens)
        arena = ALPHA[rng.randint(0, len(ALPHA), size=total)].copy() if zeros else rng.randint(0, 256, size=total).astype(np.uint8)
        idx = fake_index(lens, gaps)
        got, got_idx = ctx.emit_annexb(dev(arena), idx)
        want = orc.emit_annexb(arena, idx)
        assert np.array_equal(got, want), (mean, nn)
        pos = 0
        for k, (n, g) in enumerate(zip(lens, gaps)):
            assert int(got_idx["start"][k]) == pos + g and int(got_idx["rbsp_len"][k]) == n and int(got_idx["rbsp_off"][k]) == int(idx["rbsp_off"][k])
            pos = int(got_idx["end"][k])
        assert pos == len(want)
    arena = np.zeros(300 * 40, dtype=np.uint8)
    idx = fake_index([40] * 300, [4] * 300)
    with pytest.raises(hbs.HbsError):
        ctx.emit_annexb(dev(arena), idx, out_cap=300 * 44 + 100)      # the inserted 03s do not fit


@pytest.mark.parametrize("mean", [24, 64, 100, 160, 215])
def test_emit_tiny_nals_by_groups_of_64(ctx, orc, mean):
    """round 6: arenas of tiny NALs whose index is one stretch of the arena go through the group kernel (hbs_emit_groups.h): 64
    consecutive NALs a wavefront, the stretch staged in LDS, the output written by aligned chunks.  Random payload with a few
    patterns that need a 03 (their groups take the exact walk), NALs of every length from 0 to 2 x mean (several begin inside one
    output chunk), gaps of 3-15 bytes, the synthetic gap rule, the output index, a capacity that is too small; then the same NALs
    through an index that is NOT one stretch (every other NAL skipped: the lane per NAL) -- all against the oracle's rbsp_to_nal."""
    import hevcbitstream_amd as hbs
    rng = np.random.RandomState(600 + mean)
    nn = 30_000
    lens = [int(x) for x in rng.randint(0, 2 * mean + 1, size=nn)]
    lens[:8] = [0, 1, 2, 15, 16, 17, 0, 0]
    total = sum(lens)
    arena = rng.randint(0, 256, size=total).astype(np.uint8)
    for q in rng.randint(0, total - 8, size=40):                  # groups that need the exact walk
        arena[q:q + 3] = (0, 0, int(rng.randint(0, 4)))
    for gaps in ([int(rng.randint(3, 16)) for _ in lens], [4 if k % 4 == 0 else 3 for k in range(nn)]):
        idx = fake_index(lens, gaps)
        want = orc.emit_annexb(arena, idx)
        got, got_idx = ctx.emit_annexb(dev(arena), idx)
        assert np.array_equal(got, want), (mean, "gaps from the index")
        prev_end = np.concatenate([[0], got_idx["end"][:-1].astype(np.int64)])
        assert np.array_equal(got_idx["start"].astype(np.int64), prev_end + np.array(gaps)) and int(got_idx["end"][-1]) == len(want)
        assert np.all(got_idx["end"].astype(np.int64) - got_idx["start"].astype(np.int64) >= np.array(lens))
        for k in range(0, nn, 997):
            assert int(got_idx["rbsp_len"][k]) == lens[k] and int(got_idx["rbsp_off"][k]) == int(idx["rbsp_off"][k])
        if gaps[0] == 4 and gaps[1] == 3:
            got1, _ = ctx.emit_annexb(dev(arena), idx, gap_mode=1)
            assert np.array_equal(got1, want), (mean, "synthetic gaps")
    with pytest.raises(hbs.HbsError):
        ctx.emit_annexb(dev(arena), idx, out_cap=len(want) - 1000)
    # not one stretch: every other NAL (the arena bytes between them are nobody's)
    sub = idx[::2].copy()
    pos = 0
    for k in range(len(sub)):
        g = 3 + (k & 1)
        sub["start"][k] = pos + g
        sub["end"][k] = pos + g + int(sub["rbsp_len"][k])
        pos = int(sub["end"][k])
    # (the oracle counts the 03s a NAL takes: ends as the reference's loop would leave them are not needed for the bytes)
    want2 = orc.emit_annexb(arena, sub)
    got2, _ = ctx.emit_annexb(dev(arena), sub)
    assert np.array_equal(got2, want2), (mean, "an index with holes")


def test_emit_arena_tiles_hold_up_to_1024_nal_starts(ctx, orc):
    """round 4: a 192 KiB tile takes up to 1024 NAL starts (512 before), the ones past the first 512 fetched behind the flag
    pass -- arenas of 224-448 byte NALs go to the tile kernel instead of a lane per NAL.  Pinned to the tiles on small arenas:
    means of 300 and 230 bytes (the second passes 1024 ELEMENTS in places: those tiles are walked by rows, NAL starts in most
    of their rows), plain and zero-heavy bytes, a few NALs shorter than a chunk (several starts in one chunk); bytes against the
    oracle, the output index against the kernel by NALs, and the tile kernel did the call."""
    rng = np.random.RandomState(91)
    for mean, nn, zeros in ((300, 20000, False), (300, 12000, True), (230, 24000, False), (230, 16000, True)):
        lens = [int(x) for x in rng.randint(mean // 2, mean + mean // 2 + 1, size=nn)]
        for k in range(50, nn, 997):
            lens[k] = int(rng.randint(0, 16))
        gaps = [int(rng.randint(3, 6)) for _ in lens]
        total = sum(lens)
        arena = ALPHA[rng.randint(0, len(ALPHA), size=total)].copy() if zeros else rng.randint(0, 256, size=total).astype(np.uint8)
        idx = fake_index(lens, gaps)
        want = orc.emit_annexb(arena, idx)
        ctx.set_emit_path(0)
        _, idx_nals = ctx.emit_annexb(dev(arena), idx)
        ctx.set_emit_path(2)
        got, got_idx = ctx.emit_annexb(dev(arena), idx)
        stayed = ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h)
        ctx.set_emit_path(-1)
        assert np.array_equal(got, want), (mean, nn, zeros)
        assert np.array_equal(got_idx, idx_nals), (mean, nn, zeros)
        assert stayed == 1, (mean, nn, zeros)
    # the automatic path on an arena past the tile kernel's minimum size (192 MiB), mean 320 bytes: tiles, not a lane per NAL
    nn = 660_000
    lens = rng.randint(160, 481, size=nn).astype(np.int64)
    arena = rng.randint(0, 256, size=int(lens.sum())).astype(np.uint8)
    arena[rng.randint(0, len(arena), size=len(arena) // 50)] = 0
    idx = fake_index([int(x) for x in lens], [3 + (k & 1) for k in range(nn)])
    want = orc.emit_annexb(arena, idx)
    got, _ = ctx.emit_annexb(dev(arena), idx)
    assert ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h) == 1
    assert np.array_equal(got, want)


@pytest.mark.parametrize("mode", [0, 1])
def test_device_generator_matches_oracle(ctx, orc, mode):
    n = 3000
    stream, idx, arena = orc.gen_stream(0x1234 + mode, n, mode)
    g = ctx.synth_stream(0x1234 + mode, n, mode)
    assert g["stream_bytes"] == len(stream) and g["rbsp_bytes"] == len(arena)
    assert np.array_equal(g["rbsp"][: g["rbsp_bytes"]].cpu().numpy(), arena)
    assert np.array_equal(g["stream"][: g["stream_bytes"]].cpu().numpy(), stream)
    got_idx = g["index"].cpu().numpy().view(NAL_ENTRY)
    for f in ("start", "end", "rbsp_off", "rbsp_len"):
        assert np.array_equal(got_idx[f], idx[f]), f


def _roundtrip(ctx, n, mode, lo, hi):
    import torch
    g = ctx.synth_stream(0x1234, n, mode)
    sb, rb = g["stream_bytes"], g["rbsp_bytes"]
    assert lo < sb < hi
    stream = g["stream"][:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
    ctx.index_extract_async(stream, index, cap, rbsp, summary)
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n and int(s["rbsp_bytes"]) == rb and int(s["stop_reason"]) == -1
    assert torch.equal(rbsp[:rb], g["rbsp"][:rb])
    a = index[: n * 32].view(torch.int64).view(n, 4)
    b = g["index"][: n * 32].view(torch.int64).view(n, 4)
    assert torch.equal(a[:, :3], b[:, :3])                        # start, end, rbsp_off
    assert torch.equal(a[:-1, 3], b[:-1, 3])                      # rbsp_len|status (last NAL: UNTERMINATED flag)
    out = torch.empty(int(ctx.lib.hbs_annexb_bound(rb, n)), dtype=torch.uint8, device="cuda")
    ctx.emit_annexb_async(rbsp, rb, index, n, 0, out, None, summary)
    s2 = ctx.read_summary(summary)
    assert int(s2["error"]) == 0 and int(s2["stream_bytes"]) == sb
    assert torch.equal(out[:sb], stream)


@pytest.mark.parametrize("mode", [0, 1])
def test_roundtrip_1gib(ctx, mode):
    """configs 2 and 4 at full size, by property: generate ~1 GiB on the device, extract
    (K12), re-emit (K3) -> byte-identical stream; extracted arena == generated arena."""
    _roundtrip(ctx, 104858, mode, 1.0e9, 1.2e9)


def test_roundtrip_16gib(ctx):
    """config 5's shard at full size (what bench.py times), by the same properties: every offset past 4 GiB,
    89 478 tiles of look-back, 1 677 000 index entries"""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2**30:
        pytest.skip("needs ~70 GiB of HBM")
    _roundtrip(ctx, 1677000, 0, 17.0e9, 17.4e9)
    torch.cuda.empty_cache()


def test_gap_mode_0_with_long_zero_runs_fits_the_public_bound(orc):
    """hbs_annexb_bound_gaps: a stream with long zero runs between its NALs (trailing_zero_8bits, padding) re-emitted with the
    recorded gaps into a buffer sized by the public bound -- hbs_annexb_bound alone ignores the gaps"""
    import ctypes as C
    import torch
    import hevcbitstream_amd as hbs
    rng = np.random.RandomState(12)
    parts = []
    for k in range(200):
        parts.append(b"\x00" * int(rng.randint(0, 3000)) + b"\x00\x00\x01\x02\x01" + bytes(rng.randint(4, 256, size=int(rng.randint(1, 400))).astype(np.uint8)) + b"\x80")
    stream = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    ctx = hbs.Context(0)
    ent, arena, s = ctx.index_extract(torch.from_numpy(stream).cuda())
    assert len(ent) == 200
    gaps = int(ent["start"][0]) + int((ent["start"][1:].astype(np.int64) - ent["end"][:-1].astype(np.int64)).sum())
    lib = ctx.lib
    lib.hbs_annexb_bound_gaps.restype = C.c_uint64
    lib.hbs_annexb_bound_gaps.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    bound = int(lib.hbs_annexb_bound_gaps(len(arena), len(ent), gaps))
    assert int(lib.hbs_annexb_bound(len(arena), len(ent))) < len(stream) <= bound       # the gap-less bound would not do
    back, ent2 = ctx.emit_annexb(torch.from_numpy(arena).cuda(), ent, gap_mode=0, out_cap=bound)
    assert np.array_equal(back, stream)
    ctx.close()


def _check_emit(ctx, orc, arena, lens, gaps):
    idx = fake_index(lens, gaps)
    got, got_idx = ctx.emit_annexb(dev(arena), idx)
    want = orc.emit_annexb(arena, idx)
    assert len(got) == len(want) and np.array_equal(got, want), (len(got), len(want), lens[-4:])
    pos = 0
    for k, (n, g) in enumerate(zip(lens, gaps)):
        assert int(got_idx["start"][k]) == pos + g and int(got_idx["end"][k]) - int(got_idx["start"][k]) >= n, k
        pos = int(got_idx["end"][k])
    assert pos == len(want)


def test_emit_empty_nals_where_an_arena_of_whole_chunks_ends(path_ctx, orc):
    """Found by the long soak of round 6 (tests/tools/soak_gpu.py 1200 66, iterations 47568 and 50920): an EMPTY last NAL behind an
    arena whose length is a multiple of 16 begins in a chunk that holds no byte, and the arena-tile kernel's walk by rows (tiles in
    which every chunk of a KiB has a NAL start or a zero pair) left its start code out: the stream came out 3 / 4 bytes short.
    Every emit path, arenas of whole chunks and of whole tiles and one byte either
    side, one to three empty NALs at the end, empty NALs at chunk and tile boundaries inside, few large NALs and many small ones."""
    ctx = path_ctx
    rng = np.random.RandomState(606)
    tile = 192 * 1024
    for total in (16, 32, 1024, 44784, 52992, tile - 16, tile, tile + 16, 2 * tile, 2 * tile + 1, 2 * tile - 1, 3 * tile + 4096):
        for n_empty in (1, 2, 3):
            for many in (False, True):
                if many:                                       # several hundred small NALs: the group kernel's ground on the automatic path
                    cuts = np.sort(rng.choice(np.arange(1, total), size=min(total - 1, int(rng.randint(300, 1200))), replace=False)) if total > 1 else np.zeros(0, int)
                else:
                    cuts = np.sort(rng.choice(np.arange(1, total), size=min(total - 1, int(rng.randint(1, 6))), replace=False))
                lens = [int(x) for x in np.diff(np.concatenate(([0], cuts, [total])))]
                if total >= 2 * tile:                          # an empty NAL exactly where the second tile begins, and one at a chunk boundary
                    acc = np.cumsum(lens)
                    k = int(np.searchsorted(acc, tile))
                    lens[k:k + 1] = [lens[k] - int(acc[k] - tile), 0, int(acc[k] - tile)] if acc[k] > tile else [lens[k], 0]
                lens = lens + [0] * n_empty
                gaps = [int(rng.randint(3, 5)) for _ in lens]
                arena = ALPHA[rng.randint(0, len(ALPHA), size=total)].copy() if rng.rand() < 0.5 else rng.randint(0, 256, size=total).astype(np.uint8)
                assert sum(lens) == total
                _check_emit(ctx, orc, arena, lens, gaps)
        # tiles that are walked by rows (what the soak's streams were): every chunk of some KiB with a NAL start in it, or a KiB of zeros
        if total >= 16384:
            for shape in ("starts", "zeros"):
                head = [16] * 512 if shape == "starts" else [8192]
                rest = total - sum(head)
                lens = head + ([rest - 3000, 3000] if rest > 6000 else [rest]) + [0, 0]
                arena = rng.randint(1, 256, size=total).astype(np.uint8)
                if shape == "zeros":
                    arena[2048:5000] = 0
                _check_emit(ctx, orc, arena, lens, [3 + (k & 1) for k in range(len(lens))])


def test_emit_an_empty_last_nal_on_the_automatic_path_at_full_size(ctx, orc):
    """the same at the size from which the automatic path takes the arena tiles by itself (192 MiB): 10 KiB NALs, an arena of
    whole chunks, an empty NAL at its end -- and, as a control, the same arena one byte longer"""
    rng = np.random.RandomState(607)
    for extra in (0, 1):
        total = (200 << 20) + extra
        nn = total // 10240
        cuts = np.sort(rng.choice(np.arange(1, total, 7), size=nn - 1, replace=False))
        lens = [int(x) for x in np.diff(np.concatenate(([0], cuts, [total])))] + [0]
        gaps = [3 + (k & 1) for k in range(len(lens))]
        arena = rng.randint(0, 256, size=total).astype(np.uint8)
        arena[rng.randint(0, total, size=total // 5000)] = 0
        arena[total - 50000: total - 47000] = 0               # three KiB of zeros in the last tile: it is walked by rows
        ctx.set_emit_path(-1)
        _check_emit(ctx, orc, arena, lens, gaps)
        # the control goes by tiles (so the automatic path does take them at this size); the arena of whole chunks by NALs
        assert ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h) == extra


@pytest.mark.parametrize("case,extra", [("emit-1", 0), ("emit-1", 5), ("emit2", 0), ("scan0r", 0)])
def test_buffers_that_end_with_their_allocation(case, extra):
    """An arena of exactly 1 024 tiles (192 MiB, an allocation of its own) through the automatic emit path: until round 6 the
    arena-tile kernel's last tile -- empty, or shorter than a chunk -- read up to 192 KiB behind the arena (`room` wrapped), a GPU
    memory fault when nothing is mapped there.  In a process of its own (tests/tools/edge_faults.py: a fault kills it); the scan
    of a stream of that size beside it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "edge_faults.py"), case, str(192 << 20), str(extra)],
                       capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0 and ": ok" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-300:])
