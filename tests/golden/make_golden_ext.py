#!/usr/bin/env python3
"""Golden vectors for the opt-in extension (AUD / EOS / EOB / filler data / SEI, NAL types 35..40): every input NAL goes
through the REAL reference's never-dispatched readers (oracle/_ref/libref_ext_driver.so = oracle/ref_ext_driver.c over
libhevcref.so: read_hevc_access_unit_delimiter_rbsp, read_filler_data_rbsp, _read_ff_coded_number, read_sei_payload,
more_rbsp_data, read_hevc_rbsp_trailing_bits) and what they returned is written to tests/golden/ext_vectors.json.
Run in the dev container (needs /root/reference built by `make -C oracle`)."""
import ctypes as C
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


class RefExt(C.Structure):
    _fields_ = [("num_sei_messages", C.c_int32), ("primary_pic_type", C.c_int32), ("filler_bytes", C.c_uint32), ("reserved", C.c_uint32),
                ("sei", C.c_uint32 * 4 * 6)]


def header(t, layer=0, tid=1):
    return bytes([(t << 1) | (layer >> 5), ((layer & 31) << 3) | tid])


def ff(n):
    out = b""
    while n >= 255:
        out += b"\xff"
        n -= 255
    return out + bytes([n])


def sei_msg(rng, ptype, size, payload=None):
    if payload is None:
        payload = bytes(rng.randrange(256) for _ in range(size))
    return ff(ptype) + ff(size) + payload


def to_nal(rbsp):
    """emulation prevention as an encoder would insert it (00 00 {00..03} -> 00 00 03 xx)"""
    out = bytearray()
    z = 0
    for b in rbsp:
        if z >= 2 and b <= 3:
            out.append(3)
            z = 0
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


def cases(rng):
    c = []
    for ppt in range(8):
        for tail in (0x10, 0x1f, 0x00, 0x17):
            c.append(header(35) + bytes([(ppt << 5) | tail]))
    c += [header(35), header(35)[:1], header(35) + b"\x50\x80", header(35, layer=5, tid=3) + b"\x30", header(35) + b"\x00\x00\x03\x01\x80"]
    for t in (36, 37):
        c += [header(t), header(t) + b"\x80", header(t)[:1], header(t, tid=7) + b"\x12\x34"]
    for nff in (0, 1, 2, 7, 300):
        c += [header(38) + b"\xff" * nff + b"\x80", header(38) + b"\xff" * nff, header(38) + b"\xff" * nff + b"\x7f\x80"]
    for t in (39, 40):
        c.append(header(t) + sei_msg(rng, 5, 16) + b"\x80")
        c.append(header(t) + sei_msg(rng, 1, 3) + sei_msg(rng, 137, 24) + b"\x80")
        c.append(header(t) + sei_msg(rng, 300, 260) + b"\x80")                      # ff-coded type and size
        c.append(header(t) + b"".join(sei_msg(rng, rng.randrange(200), rng.randrange(20)) for _ in range(9)) + b"\x80")   # more than the record holds
        c.append(header(t) + sei_msg(rng, 4, 40)[:20])                               # payload runs past the NAL
        c.append(header(t) + sei_msg(rng, 6, 0) + b"\x80")
        c.append(header(t) + sei_msg(rng, 6, 2, b"\x80\x00") + b"\x80")              # a one bit that is NOT the stop bit
        c.append(header(t) + sei_msg(rng, 6, 2, b"\x12\x34") + b"\x80\x00\x00")      # trailing zero bytes (cabac_zero_words style)
        c.append(header(t) + sei_msg(rng, 0, 1, b"\x00") + b"\x00\x00\x00\x00")      # no stop bit at all: zeros read as empty messages
        c.append(header(t) + b"\xff\xff\xff")                                         # the type never ends
        c.append(header(t))
        c.append(to_nal(header(t) + sei_msg(rng, 5, 12, b"\x00\x00\x01\x00\x00\x00\x02\x00\x00\x03\x00\x00") + b"\x80"))   # emulation prevention inside
    for _ in range(120):
        t = rng.choice((35, 36, 37, 38, 39, 40))
        body = bytes(rng.choice((0, 0, 0xff, 0x80, 1, 2, 3, rng.randrange(256))) for _ in range(rng.randrange(0, 40)))
        c.append(to_nal(header(t, rng.randrange(64), rng.randrange(8)) + body))
    c += [header(32) + b"\x00", header(1) + b"\x80"]                                  # not extended types: -2
    return c


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_ext_driver.so"))
    lib.ref_read_extended_nal.argtypes = [C.c_char_p, C.c_int, C.POINTER(RefExt), C.POINTER(C.c_int)]
    rng = random.Random(20261003)
    out = []
    for nal in cases(rng):
        x = RefExt()
        t = C.c_int(0)
        rc = lib.ref_read_extended_nal(nal, len(nal), C.byref(x), C.byref(t))
        out.append({"nal": nal.hex(), "type": t.value, "rc": rc, "primary_pic_type": x.primary_pic_type, "filler_bytes": x.filler_bytes,
                    "num_sei_messages": x.num_sei_messages,
                    "sei": [[int(x.sei[i][0]), int(x.sei[i][1]), int(x.sei[i][2])] for i in range(min(6, max(0, x.num_sei_messages)))]})
    json.dump({"made_by": "tests/golden/make_golden_ext.py over oracle/_ref/libref_ext_driver.so", "vectors": out},
              open(os.path.join(HERE, "ext_vectors.json"), "w"), indent=0)
    print(len(out), "vectors")


if __name__ == "__main__":
    main()
