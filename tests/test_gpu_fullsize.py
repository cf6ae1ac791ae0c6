"""Parity at full size against the REFERENCE, not against the device generator: a bench stream of more than
5 GiB (offsets past 4 GiB) goes through the compiled reference's own loop on the host -- find_nal_unit +
nal_to_rbsp per NAL (hevc_analyze.c:135-177, h264_nal.c:38-200; oracle/_ref/libref_driver.so) -- and EVERY start /
end / rbsp_off / rbsp_len of the GPU index and EVERY byte of the GPU's RBSP arena are compared with what it
produced; then the way back: the stream K3 emits from that arena against the reference's rbsp_to_nal
(h264_nal.c:92-132) over the reference's arena.  Uniform and zero-heavy payload.  Where the prebuilt reference
did not travel with the tree, the oracle's restatement (pinned to it by tests/test_oracle_l2.py) stands in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests._refwalk import REF_ENTRY, device_equals_host, reference_emit, reference_walk      # noqa: F401


@pytest.mark.parametrize("mode,n_nals", [(0, 600_000), (1, 540_000)], ids=["uniform-5.7GiB", "zero-heavy-5.2GiB"])
def test_index_arena_and_reemission_against_the_reference(mode, n_nals):
    import torch
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    g = ctx.synth_stream(0x1234, n_nals, mode)
    sb, rb = g["stream_bytes"], g["rbsp_bytes"]
    assert sb > (5 << 30), "the stream must reach past 4 GiB offsets"
    stream = g["stream"][:sb]
    host = stream.cpu().numpy()

    # the reference's walk over the same bytes
    ent, ref_arena, ref_rb, kind = reference_walk(host, n_nals + 16)
    assert len(ent) == n_nals, (len(ent), n_nals, kind)

    # scan + index + extraction on the GPU: the default (automatic) path, then every kernel pinned
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n_nals + 16)
    for kernel in (0, 4, 6, 2):
        ctx.set_kernel(kernel)
        index.zero_()
        rbsp.zero_()
        ctx.index_extract_async(stream, index, cap, rbsp, summary)
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["nal_count"]) == n_nals and int(s["stop_reason"]) == -1, (kernel, s)
        assert int(s["rbsp_bytes"]) == ref_rb, (kernel, int(s["rbsp_bytes"]), ref_rb)
        got = index[: n_nals * 32].cpu().numpy().view(hbs.NAL_ENTRY)
        for f in ("start", "end", "rbsp_off"):
            assert np.array_equal(got[f], ent[f]), (kernel, f, int(np.flatnonzero(got[f] != ent[f])[0]))
        assert np.array_equal(got["rbsp_len"].astype(np.int64), ent["rbsp_len"].astype(np.int64)), (kernel, "rbsp_len")
        assert np.array_equal((got["status"] & hbs.ST_ERROR) != 0, ent["rc_rbsp"] < 0), (kernel, "status")
        assert int(got["start"][-1]) > (4 << 30)
        device_equals_host(rbsp, ref_arena, ref_rb, "kernel %d: RBSP arena" % kernel)
    # the index-only kernel: same entries without an arena
    ctx.set_kernel(5)
    index.zero_()
    ctx.index_extract_async(stream, index, cap, None, summary)
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n_nals
    got5 = index[: n_nals * 32].cpu().numpy().view(hbs.NAL_ENTRY)
    for f in ("start", "end"):
        assert np.array_equal(got5[f], ent[f]), (5, f)
    ctx.set_kernel(0)
    del host

    # the way back: K3 over the GPU's arena against the reference's rbsp_to_nal over the reference's arena
    ref_stream, ref_sb = reference_emit(ref_arena, ent, sb + (1 << 16))      # room for the 3/2 bound of the last NAL
    assert ref_sb == sb, (ref_sb, sb)
    out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
    idx_out = torch.empty(n_nals * 32, dtype=torch.uint8, device="cuda")
    dev_index = torch.from_numpy(got.view(np.uint8).copy()).cuda()
    for path in (0, 1, 2):                   # single pass (by items), three-step, arena tiles
        ctx.set_emit_path(path)
        out.zero_()
        ctx.emit_annexb_async(rbsp, ref_rb, dev_index, n_nals, 1, out, idx_out, summary)
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["stream_bytes"]) == sb, (path, s)
        device_equals_host(out, ref_stream, sb, "emit path %d: re-emitted stream" % path)
    ctx.set_emit_path(-1)
    ctx.close()
