"""Outputs of exactly the capacity the caller states, with canary bytes behind them (dev aid, round 6): a kernel that stores past
the capacity it was given overwrites the caller's memory without anybody noticing -- here the canaries notice.  Scan + extraction
(every kernel; the arena's capacity = the RBSP bytes, + 0 / + 16; the index's = the NALs found) and RBSP -> Annex-B (every path;
the output's capacity = the bytes that come out), on streams of small and large NALs, plain and zero-heavy, sizes around the tiles.
A call may answer HBS_E_CAPACITY instead of fitting -- a stream that stops at an empty NAL (stop_reason 1) is judged by the NALs and
RBSP bytes FOUND, the ones behind the stop included (hbs_summary.nal_found; include/hevcbitstream_amd.h) -- those are counted and
listed at the end; what may never happen is a changed canary or, without an error, a result that differs from the oracle's.
usage: python3 tests/tools/canaries.py [seconds] [seed]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from hevcbitstream_amd.api import NAL_ENTRY, SUMMARY
from tests import _orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = _orc.oracle()
CAN = 0xC3
PAD = 1 << 16


def canary(nbytes):
    return torch.full((nbytes + PAD,), CAN, dtype=torch.uint8, device="cuda")


def intact(t, used):
    return bool((t[used:] == CAN).all().item())


def make_stream(rng):
    size = int(rng.choice([1000, 65536, 98304, 196608, 196608 * 2, 262144, 300000, 1 << 20, 3 * 196608 + 5, 5_000_000]))
    mean = int(rng.choice([40, 300, 2000, 9000, 100000]))
    s = rng.integers(1, 256, size=size, dtype=np.uint8)
    kind = int(rng.integers(0, 3))
    if kind == 1:
        s[rng.random(size) < 0.15] = 0
    elif kind == 2:
        s[rng.random(size) < 0.01] = 0
    pos = 0
    while pos + 8 < size:
        s[pos:pos + 4] = (0, 0, 1, 0x42) if rng.random() < 0.5 else (0, 0, 0, 1)
        pos += int(rng.integers(max(4, mean // 2), mean * 3 // 2 + 5))
    return s


ctxs = {}
for v in (0, 2, 4, 5, 6):
    ctxs[v] = hbs.Context(0)
    ctxs[v].set_kernel(v)
emit = {}
for p in (-1, 0, 1, 2):
    emit[p] = hbs.Context(0)
    emit[p].set_emit_path(p)
t_end = time.time() + budget
it = bad = 0
notes = {}
while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 50021 + it)
    s = make_stream(rng)
    want_idx, want_arena, why = orc.index_extract(s)
    n = len(want_idx)
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if n else 0
    d = torch.from_numpy(s).cuda()
    for v, c in ctxs.items():
        for slack in (0, 16):
            if v == 5 and slack:
                continue
            cap_n = max(n, 1)
            index = canary(cap_n * NAL_ENTRY.itemsize)
            rbsp = canary(tot + slack) if v != 5 else None
            summ = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
            try:
                c.index_extract_async(d, index[: cap_n * NAL_ENTRY.itemsize], cap_n, rbsp[: tot + slack] if rbsp is not None else None, summ)
                sm = c.read_summary(summ)
                err = int(sm["error"])
            except hbs.HbsError:                 # refused on the host (a capacity of 0, ...): nothing was launched
                sm, err = None, -3
            ok_can = intact(index, cap_n * NAL_ENTRY.itemsize) and (rbsp is None or intact(rbsp, tot + slack))
            if not ok_can:
                bad += 1
                print("CANARY scan kernel", v, "slack", slack, "iter", it, "len", len(s), "nals", n, "rbsp", tot, "error", err, flush=True)
            if err == 0:
                got = index[: n * NAL_ENTRY.itemsize].cpu().numpy().view(NAL_ENTRY)
                same = int(sm["nal_count"]) == n and all(np.array_equal(got[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status"))
                if rbsp is not None:
                    same = same and np.array_equal(rbsp[:tot].cpu().numpy(), want_arena[:tot])
                if not same:
                    bad += 1
                    print("MISMATCH scan kernel", v, "slack", slack, "iter", it, "len", len(s), flush=True)
            else:
                notes[("scan", v, slack, err)] = notes.get(("scan", v, slack, err), 0) + 1
    keep = want_idx[(want_idx["status"] & 1) == 0]
    if len(keep):
        want = orc.emit_annexb(want_arena, keep)
        d_idx = torch.from_numpy(np.ascontiguousarray(keep).view(np.uint8).copy()).cuda()
        # the arena and the output at any byte alignment (the scan asks for 16-byte aligned pointers; hbs_emit_annexb does not)
        oa, oo = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        arena_buf = torch.empty(len(want_arena) + 32, dtype=torch.uint8, device="cuda")
        arena = arena_buf[oa: oa + len(want_arena)]
        arena.copy_(torch.from_numpy(want_arena.copy()))
        for p, c in emit.items():
            for slack in (0, 16):
                out_buf = canary(len(want) + slack + 16)
                out = out_buf[oo:]
                summ = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
                try:
                    c.emit_annexb_async(arena, int(arena.numel()), d_idx, len(keep), 0, out[: len(want) + slack], None, summ)
                    sm = c.read_summary(summ)
                    err = int(sm["error"])
                except hbs.HbsError:
                    sm, err = None, -3
                if not (intact(out, len(want) + slack) and bool((out_buf[:oo] == CAN).all().item())):
                    bad += 1
                    print("CANARY emit path", p, "offsets", oa, oo, "slack", slack, "iter", it, "arena", int(arena.numel()), "nals", len(keep), "out", len(want), "error", err, flush=True)
                if err == 0:
                    if int(sm["stream_bytes"]) != len(want) or not np.array_equal(out[: len(want)].cpu().numpy(), want):
                        bad += 1
                        print("MISMATCH emit path", p, "slack", slack, "iter", it, flush=True)
                else:
                    notes[("emit", p, slack, err)] = notes.get(("emit", p, slack, err), 0) + 1
    it += 1
print("calls that reported an error instead of fitting (kind, kernel / path, slack, error): count", sorted(notes.items()))
print("iterations", it, "bad", bad)
