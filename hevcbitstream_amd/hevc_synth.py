"""hevc_synth.py -- test-side generator of HEVC header NAL units.

There is no real HEVC stream in the reference tree and no encoder in the
image, so the parse tests and config 3 ("4K30 stream") are fed by this writer.
It walks the syntax in the order the reference READER consumes it
(hevc_stream.c:243-1218, including its departures from H.265),
picks a value for every element inside the envelope where the reference is
defined (ids 0, <= 31 short-term RPS, <= 32 entry points, ...), and emits the
bits.  It tracks the parameter-set fields and the derived RPS tables the slice
reader depends on, exactly as the reader would have them.

Workload generator for the tests and for bench.py (BASELINE config 3); not part of the C-ABI product."""
import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        v = int(v)
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def u1(self, v):
        self.bits.append(int(v) & 1)

    def ue(self, v):
        v = int(v) + 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v):
        v = int(v)
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def aligned(self):
        return len(self.bits) % 8 == 0

    def trailing(self):
        """what read_hevc_rbsp_trailing_bits / byte_alignment skip: 1 then zeros to the boundary"""
        self.u1(1)
        while not self.aligned():
            self.u1(0)

    def bytes(self):
        b = list(self.bits)
        while len(b) % 8:
            b.append(0)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def rbsp_to_nal(rbsp):
    out = bytearray()
    zeros = 0
    for v in rbsp:
        if zeros == 2 and v <= 3:
            out.append(3)
            zeros = 0
        out.append(v)
        zeros = zeros + 1 if v == 0 else 0
    return bytes(out)


def ceil_log2(n):
    return 0 if n <= 1 else (n - 1).bit_length()


class Synth:
    """Generates NALs and mirrors the reader's state (last SPS/PPS, RPS tables)."""

    def __init__(self, seed=0, rich=True):
        self.rng = np.random.RandomState(seed)
        self.rich = rich                      # rich: exercise rare branches; plain: x265-like headers
        self.sps = None
        self.pps = None
        # derived RPS tables of the reader (hevc_stream.c:26-32)
        self.NumDeltaPocs = [0] * 33
        self.NumNeg = [0] * 33
        self.NumPos = [0] * 33
        self.DeltaPocS0 = [[0] * 32 for _ in range(33)]
        self.UsedS0 = [[0] * 32 for _ in range(33)]
        self.DeltaPocS1 = [[0] * 32 for _ in range(33)]
        self.UsedS1 = [[0] * 32 for _ in range(33)]

    # -- helpers ---------------------------------------------------------------------
    def chance(self, p):
        return self.rich and self.rng.rand() < p

    def ri(self, lo, hi):
        return int(self.rng.randint(lo, hi + 1))

    def nal(self, nal_type, w, layer=0, tid=1):
        hdr = BitWriter()
        hdr.u1(0)
        hdr.u(6, nal_type)
        hdr.u(6, layer)
        hdr.u(3, tid)
        return rbsp_to_nal(hdr.bytes() + w.bytes())

    # -- 7.3.3 as read by hevc_stream.c:652-755 ----------------------------------------
    def ptl(self, w, max_sub_layers_minus1):
        idc = self.ri(1, 2) if not self.rich else int(self.rng.choice([1, 2, 3, 4, 5, 9]))
        w.u(2, 0)
        w.u1(self.ri(0, 1))
        w.u(5, idc)
        compat = [0] * 32
        compat[idc if idc < 32 else 1] = 1
        if self.chance(0.3):
            compat[self.ri(1, 7)] = 1
        for i in range(32):
            w.u1(compat[i])
        for _ in range(4):
            w.u1(self.ri(0, 1))
        if idc in (4, 5, 6, 7) or compat[4] or compat[5] or compat[6] or compat[7]:
            for _ in range(9):
                w.u1(self.ri(0, 1))
            w.u(34, 0)
        else:
            w.u(43, 0)
        w.u1(self.ri(0, 1))                 # general_inbld_flag or reserved bit: one bit either way
        w.u(8, int(self.rng.choice([93, 120, 123, 150, 153])))
        present = []
        for i in range(max_sub_layers_minus1):
            p, l = self.ri(0, 1), self.ri(0, 1)
            present.append((p, l))
            w.u1(p)
            w.u1(l)
        if max_sub_layers_minus1 > 0:
            for i in range(max_sub_layers_minus1, 8):
                w.u(2, 0)
        for i in range(max_sub_layers_minus1):
            p, l = present[i]
            if p:
                sidc = int(self.rng.choice([1, 2, 4, 9]))
                w.u(2, 0)
                w.u1(self.ri(0, 1))
                w.u(5, sidc)
                sc = [0] * 32
                sc[sidc] = 1
                for j in range(32):
                    w.u1(sc[j])
                for _ in range(4):
                    w.u1(self.ri(0, 1))
                if sidc in (4, 5, 6, 7) or sc[4] or sc[5] or sc[6] or sc[7]:
                    for _ in range(9):
                        w.u1(self.ri(0, 1))
                    w.u(34, 0)
                else:
                    w.u(43, 0)
                w.u1(self.ri(0, 1))          # sub_layer_inbld_flag: always read (:739-745)
            if l:
                w.u(8, self.ri(30, 186))

    # -- E.2.2 / E.2.3 as read by :1160-1218 ---------------------------------------------
    def hrd(self, w, common, max_sub_layers_minus1):
        nal_p = vcl_p = sub_pic = 0
        if common:
            nal_p, vcl_p = self.ri(0, 1), self.ri(0, 1)
            w.u1(nal_p)
            w.u1(vcl_p)
            if nal_p or vcl_p:
                sub_pic = self.ri(0, 1)
                w.u1(sub_pic)
                if sub_pic:
                    w.u(8, self.ri(0, 255))
                    w.u(5, self.ri(0, 31))
                    w.u1(self.ri(0, 1))
                    w.u(5, self.ri(0, 31))
                w.u(4, self.ri(0, 15))
                w.u(4, self.ri(0, 15))
                if sub_pic:
                    w.u(4, self.ri(0, 15))
                w.u(5, self.ri(0, 31))
                w.u(5, self.ri(0, 31))
                w.u(5, self.ri(0, 31))
        for i in range(max_sub_layers_minus1 + 1):
            general = self.ri(0, 1)
            w.u1(general)
            within = 0
            if not general:
                within = self.ri(0, 1)
                w.u1(within)
            low_delay = 0
            if within:
                w.ue(self.ri(0, 100))
            else:
                low_delay = self.ri(0, 1)
                w.u1(low_delay)
            cpb_cnt_minus1 = 0
            if low_delay:
                cpb_cnt_minus1 = self.ri(0, 3)
                w.ue(cpb_cnt_minus1)
            for present in (nal_p, vcl_p):
                if present:
                    for _ in range(cpb_cnt_minus1 + 2):          # i <= CpbCnt, CpbCnt = cnt_minus1 + 1
                        w.ue(self.ri(0, 5000))
                        w.ue(self.ri(0, 5000))
                        if sub_pic:
                            w.ue(self.ri(0, 5000))
                            w.ue(self.ri(0, 5000))
                        w.u1(self.ri(0, 1))

    # -- 7.3.4 as read by :758-779 -----------------------------------------------------------
    def scaling_list(self, w):
        for size_id in range(4):
            for matrix_id in range(0, 6, 3 if size_id == 3 else 1):
                mode = self.ri(0, 1)
                w.u1(mode)
                if not mode:
                    w.ue(self.ri(0, matrix_id if size_id < 3 else matrix_id // 3))
                else:
                    if size_id > 1:
                        w.se(self.ri(-7, 40))
                    for _ in range(min(64, 1 << (4 + (size_id << 1)))):
                        w.se(self.ri(-20, 20))

    # -- 7.3.7 as read by :1032-1085, with the derivation of :61-113 -----------------------------
    def st_rps(self, w, idx, num_sets):
        inter = 0
        if idx != 0:
            inter = 1 if self.chance(0.4) else 0
            w.u1(inter)
        if inter:
            delta_idx_minus1 = 0
            if idx == num_sets:
                delta_idx_minus1 = self.ri(0, min(idx - 1, 2))
                w.ue(delta_idx_minus1)
            sign = self.ri(0, 1)
            absd = self.ri(0, 3)
            w.u1(sign)
            w.ue(absd)
            ref = idx - (delta_idx_minus1 + 1)
            used, use_delta = [0] * 40, [0] * 40
            for j in range(self.NumDeltaPocs[ref] + 1):
                used[j] = self.ri(0, 1)
                w.u1(used[j])
                if not used[j]:
                    use_delta[j] = self.ri(0, 1)
                    w.u1(use_delta[j])
            # updateNumDeltaPocs (:61-113)
            d_rps = (1 - 2 * sign) * (absd + 1)
            i = 0
            for j in range(self.NumPos[ref] - 1, -1, -1):
                d = self.DeltaPocS1[ref][j] + d_rps
                if d < 0 and use_delta[self.NumNeg[ref] + j]:
                    self.DeltaPocS0[idx][i] = d
                    self.UsedS0[idx][i] = used[self.NumNeg[ref] + j]
                    i += 1
            if d_rps < 0 and use_delta[self.NumDeltaPocs[ref]]:
                self.DeltaPocS0[idx][i] = d_rps
                self.UsedS0[idx][i] = used[self.NumDeltaPocs[ref]]
                i += 1
            for j in range(self.NumNeg[ref]):
                d = self.DeltaPocS0[ref][j] + d_rps
                if d < 0 and use_delta[j]:
                    self.DeltaPocS0[idx][i] = d
                    self.UsedS0[idx][i] = used[j]
                    i += 1
            self.NumNeg[idx] = i
            i = 0
            for j in range(self.NumNeg[ref] - 1, -1, -1):
                d = self.DeltaPocS0[ref][j] + d_rps
                if d > 0 and use_delta[j]:
                    self.DeltaPocS1[idx][i] = d
                    self.UsedS1[idx][i] = used[j]
                    i += 1
            if d_rps > 0 and use_delta[self.NumDeltaPocs[ref]]:
                self.DeltaPocS1[idx][i] = d_rps
                self.UsedS1[idx][i] = used[self.NumDeltaPocs[ref]]
                i += 1
            for j in range(self.NumPos[ref]):
                d = self.DeltaPocS1[ref][j] + d_rps
                if d > 0 and use_delta[self.NumNeg[ref] + j]:
                    self.DeltaPocS1[idx][i] = d
                    self.UsedS1[idx][i] = used[self.NumNeg[ref] + j]
                    i += 1
            self.NumPos[idx] = i
        else:
            neg, pos = self.ri(0, 4), self.ri(0, 3)
            w.ue(neg)
            w.ue(pos)
            acc = 0
            for i in range(neg):
                d = self.ri(0, 3)
                u = self.ri(0, 1)
                w.ue(d)
                w.u1(u)
                acc -= d + 1
                self.DeltaPocS0[idx][i] = acc
                self.UsedS0[idx][i] = u
            acc = 0
            for i in range(pos):
                d = self.ri(0, 3)
                u = self.ri(0, 1)
                w.ue(d)
                w.u1(u)
                acc += d + 1
                self.DeltaPocS1[idx][i] = acc
                self.UsedS1[idx][i] = u
            self.NumNeg[idx] = neg
            self.NumPos[idx] = pos
        self.NumDeltaPocs[idx] = self.NumNeg[idx] + self.NumPos[idx]

    # -- 7.3.2.1 as read by :243-300 ----------------------------------------------------------------
    def vps(self):
        w = BitWriter()
        msl = self.ri(0, 2) if self.rich else 0
        w.u(4, self.ri(0, 15) if self.rich else 0)
        w.u1(1)
        w.u1(1)
        w.u(6, 0)
        w.u(3, msl)
        w.u1(1)
        w.u(16, 0xFFFF)
        self.ptl(w, msl)
        info = self.ri(0, 1)
        w.u1(info)
        for i in range(0 if info else msl, msl + 1):
            w.ue(self.ri(0, 6))
            w.ue(self.ri(0, 4))
            w.ue(self.ri(0, 8))
        max_layer_id = self.ri(0, 3) if self.rich else 0
        w.u(6, max_layer_id)
        sets = self.ri(0, 2) if self.rich else 0
        w.ue(sets)
        for i in range(1, sets + 1):
            for j in range(max_layer_id + 1):
                w.u1(self.ri(0, 1))
        timing = 1 if self.chance(0.6) else 0
        w.u1(timing)
        if timing:
            w.u(32, self.ri(1, 2 ** 31 - 1) if not self.chance(0.2) else 0xC0000001)
            w.u(32, self.ri(1, 2 ** 31 - 1))
            poc = self.ri(0, 1)
            w.u1(poc)
            if poc:
                w.ue(self.ri(0, 10))
            nhrd = self.ri(0, 2)
            w.ue(nhrd)
            for i in range(nhrd):
                w.ue(self.ri(0, sets))
                cprms = 0                      # cprms_present_flag[0] stays 0 (memset), :287-292
                if i > 0:
                    cprms = self.ri(0, 1)
                    w.u1(cprms)
                self.hrd(w, cprms, msl)
        w.u1(0)
        w.trailing()
        return self.nal(32, w)

    # -- 7.3.2.2 as read by :303-401 -------------------------------------------------------------------
    def sps_nal(self, width=1920, height=1080, ctb_log2=None, force=None):
        force = force or {}
        w = BitWriter()
        s = {}
        msl = self.ri(0, 2) if self.rich else 0
        w.u(4, 0)
        w.u(3, msl)
        w.u1(1)
        self.ptl(w, msl)
        w.ue(0)                                 # sps_seq_parameter_set_id: 0 (envelope)
        chroma = force.get("chroma_format_idc", int(self.rng.choice([0, 1, 1, 1, 2, 3])) if self.rich else 1)
        w.ue(chroma)
        sep = 0
        if chroma == 3:
            sep = self.ri(0, 1)
            w.u1(sep)
        w.ue(width)
        w.ue(height)
        cw = 1 if self.chance(0.4) else 0
        w.u1(cw)
        if cw:
            for _ in range(4):
                w.ue(self.ri(0, 8))
        w.ue(self.ri(0, 2) if self.rich else 0)
        w.ue(self.ri(0, 2) if self.rich else 0)
        poc_bits_minus4 = self.ri(0, 8)
        w.ue(poc_bits_minus4)
        info = self.ri(0, 1)
        w.u1(info)
        for i in range(0 if info else msl, msl + 1):
            w.ue(self.ri(0, 6))
            w.ue(self.ri(0, 4))
            w.ue(self.ri(0, 8))
        min_cb = self.ri(0, 1)
        diff = (ctb_log2 - 3 - min_cb) if ctb_log2 else self.ri(1, 3 - min_cb)
        w.ue(min_cb)
        w.ue(diff)
        w.ue(0)
        w.ue(self.ri(0, 3))
        w.ue(self.ri(0, 3))
        w.ue(self.ri(0, 3))
        sl = 1 if self.chance(0.3) else 0
        w.u1(sl)
        if sl:
            present = self.ri(0, 1)
            w.u1(present)
            if present:
                self.scaling_list(w)
        w.u1(self.ri(0, 1))
        sao = self.ri(0, 1) if self.rich else 1
        w.u1(sao)
        pcm = 1 if self.chance(0.3) else 0
        w.u1(pcm)
        if pcm:
            w.u(4, self.ri(0, 7))
            w.u(4, self.ri(0, 7))
            w.ue(self.ri(0, 2))
            w.ue(self.ri(0, 2))
            w.u1(self.ri(0, 1))
        nsets = force.get("num_short_term_ref_pic_sets", self.ri(0, 6) if self.rich else self.ri(1, 4))
        w.ue(nsets)
        for i in range(nsets):
            self.st_rps(w, i, nsets)
        lt = 1 if self.chance(0.4) else 0
        w.u1(lt)
        nlt = 0
        lt_used = []
        if lt:
            nlt = self.ri(0, 4)
            w.ue(nlt)
            for i in range(nlt):
                w.u(poc_bits_minus4 + 4, self.ri(0, (1 << (poc_bits_minus4 + 4)) - 1))
                u = self.ri(0, 1)
                lt_used.append(u)
                w.u1(u)
        tmvp = self.ri(0, 1)
        w.u1(tmvp)
        w.u1(self.ri(0, 1))
        vui = 1 if self.chance(0.5) else 0
        w.u1(vui)
        if vui:
            self.vui(w, msl)
        ext = 1 if self.chance(0.3) else 0
        w.u1(ext)
        range_ext = 0
        if ext:
            range_ext = self.ri(0, 1)
            w.u1(range_ext)
            w.u1(0)
            w.u1(0)
            w.u(5, 0)
        if range_ext:
            for _ in range(9):
                w.u1(self.ri(0, 1))
        # the reference reads no rbsp_trailing_bits here; a real SPS still carries them
        w.trailing()
        s.update(chroma_format_idc=chroma, separate_colour_plane_flag=sep, width=width, height=height,
                 log2_min_cb_minus3=min_cb, log2_diff=diff, poc_bits=poc_bits_minus4 + 4, sao=sao,
                 num_sets=nsets, lt_present=lt, num_lt_sps=nlt, lt_used=lt_used, tmvp=tmvp)
        self.sps = s
        return self.nal(33, w)

    # -- E.2.1 as read by :1088-1157 -------------------------------------------------------------------------
    def vui(self, w, msl):
        ar = self.ri(0, 1)
        w.u1(ar)
        if ar:
            idc = int(self.rng.choice([1, 2, 255]))
            w.u(8, idc)
            if idc == 255:
                w.u(16, self.ri(1, 65535))
                w.u(16, self.ri(1, 65535))
        ov = self.ri(0, 1)
        w.u1(ov)
        if ov:
            w.u1(self.ri(0, 1))
        vs = self.ri(0, 1)
        w.u1(vs)
        if vs:
            w.u(3, self.ri(0, 5))
            w.u1(self.ri(0, 1))
            cd = self.ri(0, 1)
            w.u1(cd)
            if cd:
                w.u(8, self.ri(1, 9))
                w.u(8, self.ri(1, 18))
                w.u(8, self.ri(0, 9))
        cl = self.ri(0, 1)
        w.u1(cl)
        if cl:
            w.ue(self.ri(0, 5))
            w.ue(self.ri(0, 5))
        w.u1(self.ri(0, 1))
        w.u1(self.ri(0, 1))
        w.u1(self.ri(0, 1))
        dd = self.ri(0, 1)
        w.u1(dd)
        if dd:
            for _ in range(4):
                w.ue(self.ri(0, 16))
        ti = self.ri(0, 1)
        w.u1(ti)
        if ti:
            w.u(32, self.ri(1, 100000))
            w.u(32, self.ri(1, 100000))
            poc = self.ri(0, 1)
            w.u1(poc)
            if poc:
                w.ue(self.ri(0, 10))
            hp = self.ri(0, 1)
            w.u1(hp)
            if hp:
                self.hrd(w, 1, msl)
        br = self.ri(0, 1)
        w.u1(br)
        if br:
            w.u1(self.ri(0, 1))
            w.u1(self.ri(0, 1))
            w.u1(self.ri(0, 1))
            w.ue(self.ri(0, 100))
            w.ue(self.ri(0, 16))
            w.ue(self.ri(0, 16))
            w.ue(self.ri(0, 15))
            w.ue(self.ri(0, 15))

    # -- 7.3.2.3 as read by :419-521 ------------------------------------------------------------------------------
    def pps_nal(self, force=None):
        force = force or {}
        w = BitWriter()
        p = {}
        w.ue(0)                                 # pic_parameter_set_id: 0 (envelope)
        w.ue(0)                                 # seq_parameter_set_id: 0
        dep = force.get("dependent", self.ri(0, 1) if self.rich else 0)
        w.u1(dep)
        outp = self.ri(0, 1) if self.rich else 0
        w.u1(outp)
        extra = self.ri(0, 2) if self.rich else 0
        w.u(3, extra)
        w.u1(self.ri(0, 1))
        cabac_init = self.ri(0, 1)
        w.u1(cabac_init)
        l0 = self.ri(0, 3)
        l1 = self.ri(0, 3)
        w.ue(l0)
        w.ue(l1)
        w.se(self.ri(-10, 10))
        w.u1(self.ri(0, 1))
        ts = self.ri(0, 1)
        w.u1(ts)
        cuqp = self.ri(0, 1)
        w.u1(cuqp)
        if cuqp:
            w.ue(self.ri(0, 3))
        w.se(self.ri(-6, 6))
        w.se(self.ri(-6, 6))
        chroma_off = self.ri(0, 1)
        w.u1(chroma_off)
        wp, wbp = self.ri(0, 1), self.ri(0, 1)
        w.u1(wp)
        w.u1(wbp)
        w.u1(self.ri(0, 1))
        tiles = force.get("tiles", 1 if self.chance(0.3) else 0)
        w.u1(tiles)
        wpp = force.get("wpp", self.ri(0, 1))
        w.u1(wpp)
        if tiles:
            cols, rows = self.ri(0, 3), self.ri(0, 3)
            w.ue(cols)
            w.ue(rows)
            uni = self.ri(0, 1)
            w.u1(uni)
            if not uni:
                for _ in range(cols):
                    w.ue(self.ri(0, 5))
                for _ in range(rows):
                    w.ue(self.ri(0, 5))
            w.u1(self.ri(0, 1))
        lf_slices = self.ri(0, 1)
        w.u1(lf_slices)
        dbc = self.ri(0, 1)
        w.u1(dbc)
        override = 0
        if dbc:
            override = self.ri(0, 1)
            w.u1(override)
            dis = self.ri(0, 1)
            w.u1(dis)
            if dis:                              # :471: offsets are read when the flag is 1
                w.se(self.ri(-6, 6))
                w.se(self.ri(-6, 6))
        sl = 1 if self.chance(0.2) else 0
        w.u1(sl)
        if sl:
            self.scaling_list(w)
        lm = force.get("lists_mod", self.ri(0, 1))
        w.u1(lm)
        w.ue(self.ri(0, 3))
        she = 1 if self.chance(0.3) else 0
        w.u1(she)
        ext = 1 if self.chance(0.3) else 0
        w.u1(ext)
        range_ext = 0
        if ext:
            range_ext = self.ri(0, 1)
            w.u1(range_ext)
            w.u1(0)
            w.u1(0)
            w.u1(0)                              # pps_extension_5bits read as ONE bit (:488)
        cqo_list = 0
        if range_ext:
            if ts:
                w.ue(self.ri(0, 3))
            w.u1(self.ri(0, 1))
            cqo_list = self.ri(0, 1)
            w.u1(cqo_list)
            if cqo_list:
                w.ue(self.ri(0, 2))
                n = self.ri(0, 5)
                w.ue(n)
                for _ in range(n + 1):
                    w.se(self.ri(-12, 12))
                    w.se(self.ri(-12, 12))
            w.ue(self.ri(0, 2))
            w.ue(self.ri(0, 2))
        w.trailing()
        p.update(dependent=dep, output_flag_present=outp, extra_bits=extra, cabac_init_present=cabac_init,
                 l0=l0, l1=l1, chroma_offsets=chroma_off, weighted=wp, weighted_bi=wbp, tiles=tiles, wpp=wpp,
                 lf_across_slices=lf_slices, deblock_override=override, lists_mod=lm, header_ext=she,
                 cqo_list=cqo_list)
        self.pps = p
        return self.nal(34, w)

    # -- 7.3.6 as read by :782-1029 ------------------------------------------------------------------------------
    def _num_pic_total_curr(self, sps_flag, rps_idx, lt_used_flags):
        cur = rps_idx if sps_flag else self.sps["num_sets"]
        n = sum(1 for i in range(self.NumNeg[cur]) if self.UsedS0[cur][i])
        n += sum(1 for i in range(self.NumPos[cur]) if self.UsedS1[cur][i])
        n += sum(1 for u in lt_used_flags if u)
        return n

    def slice_nal(self, nal_type, first=True, payload=b"", slice_type=None, address=0, tid=1, pps_id=0):
        s, p = self.sps, self.pps
        w = BitWriter()
        w.u1(1 if first else 0)
        if 16 <= nal_type <= 23:
            w.u1(self.ri(0, 1))
        w.ue(pps_id)                            # slice_pic_parameter_set_id: 0 (envelope); anything else makes the reference read the
        #                                         header against its all-zero sets -- and, if it codes an own RPS, write it into row 0 of the tables
        dependent = 0
        if not first:
            if p["dependent"]:
                dependent = 1 if self.chance(0.3) else 0
                w.u1(dependent)
            ctb = 1 << (s["log2_min_cb_minus3"] + 3 + s["log2_diff"])
            n_ctb = -(-s["width"] // ctb) * -(-s["height"] // ctb)
            bits = ceil_log2(n_ctb)
            w.u(bits, address % max(1, min(n_ctb, 1 << bits)) if bits else 0)
        if not dependent:
            for _ in range(p["extra_bits"]):
                w.u1(self.ri(0, 1))
            idr = nal_type in (19, 20)
            if slice_type is None:
                slice_type = 2 if idr else int(self.rng.choice([0, 1, 2]))
            w.ue(slice_type)
            if p["output_flag_present"]:
                w.u1(self.ri(0, 1))
            if s["separate_colour_plane_flag"] == 1:
                w.u(2, self.ri(0, 2))
            sps_flag, rps_idx = 0, 0
            lt_used_flags = []
            tmvp_slice = 0
            if not idr:
                w.u(s["poc_bits"], self.ri(0, (1 << s["poc_bits"]) - 1))
                sps_flag = 1 if (s["num_sets"] > 0 and self.rng.rand() < 0.6) else 0
                w.u1(sps_flag)
                if not sps_flag:
                    self.st_rps(w, s["num_sets"], s["num_sets"])
                elif s["num_sets"] > 1:
                    rps_idx = self.ri(0, s["num_sets"] - 1)
                    w.u(ceil_log2(s["num_sets"]), rps_idx)
                if s["lt_present"]:
                    n_lt_sps = 0
                    if s["num_lt_sps"] > 0:
                        n_lt_sps = self.ri(0, min(2, s["num_lt_sps"]))
                        w.ue(n_lt_sps)
                    n_lt = self.ri(0, 2)
                    w.ue(n_lt)
                    for i in range(n_lt_sps + n_lt):
                        if i < n_lt_sps:
                            k = 0
                            if s["num_lt_sps"] > 1:
                                k = self.ri(0, s["num_lt_sps"] - 1)
                                w.u(ceil_log2(s["num_lt_sps"]), k)
                            lt_used_flags.append(s["lt_used"][k])
                        else:
                            w.u(s["poc_bits"], self.ri(0, (1 << s["poc_bits"]) - 1))
                            u = self.ri(0, 1)
                            w.u1(u)
                            lt_used_flags.append(u)
                        msb = self.ri(0, 1)
                        w.u1(msb)
                        if msb:
                            w.ue(self.ri(0, 4))
                if s["tmvp"]:
                    tmvp_slice = self.ri(0, 1)
                    w.u1(tmvp_slice)
            sao_l = sao_c = 0
            if s["sao"]:
                sao_l = self.ri(0, 1)
                w.u1(sao_l)
                cat = s["chroma_format_idc"] if s["separate_colour_plane_flag"] == 0 else 0
                if cat != 0:
                    sao_c = self.ri(0, 1)
                    w.u1(sao_c)
            if slice_type in (0, 1):
                l0, l1 = p["l0"], p["l1"]
                ov = self.ri(0, 1)
                w.u1(ov)
                if ov:
                    l0 = self.ri(0, 4)
                    w.ue(l0)
                    if slice_type == 0:
                        l1 = self.ri(0, 4)
                        w.ue(l1)
                if p["lists_mod"]:
                    # getNumPicTotalCurr (:35-59) indexes used_by_curr_pic_lt_flag[i] with the loop
                    # index for the slice's own long-term pictures: the generator mirrors that
                    npc = self._num_pic_total_curr(sps_flag, rps_idx, lt_used_flags)
                    if npc > 1:
                        m0 = self.ri(0, 1)
                        w.u1(m0)
                        if m0:
                            for _ in range(l0 + 1):
                                w.u(ceil_log2(npc), self.ri(0, npc - 1))
                        # list1's flag is never read (:959)
                if slice_type == 0:
                    w.u1(self.ri(0, 1))
                if p["cabac_init_present"]:
                    w.u1(self.ri(0, 1))
                if tmvp_slice:
                    col_l0 = 1
                    if slice_type == 0:
                        col_l0 = self.ri(0, 1)
                        w.u1(col_l0)
                    if (col_l0 and l0 > 0) or (not col_l0 and l1 > 0):
                        w.ue(self.ri(0, l0 if col_l0 else l1))
                if (p["weighted"] and slice_type == 1) or (p["weighted_bi"] and slice_type == 0):
                    self.pwt(w, slice_type, l0, l1)
                w.ue(self.ri(0, 4))
            w.se(self.ri(-12, 12))
            if p["chroma_offsets"]:
                w.se(self.ri(-6, 6))
                w.se(self.ri(-6, 6))
            if p["cqo_list"]:
                w.u1(self.ri(0, 1))
            dbo = 0
            if p["deblock_override"]:
                dbo = self.ri(0, 1)
                w.u1(dbo)
            dis = 0
            if dbo:
                dis = self.ri(0, 1)
                w.u1(dis)
                if not dis:
                    w.se(self.ri(-6, 6))
                    w.se(self.ri(-6, 6))
            if p["lf_across_slices"] and (sao_l or sao_c or not dis):
                w.u1(self.ri(0, 1))
        if p["tiles"] or p["wpp"]:
            n = self.ri(0, 32) if self.rich else self.ri(0, 8)
            w.ue(n)
            if n > 0:
                ol = self.ri(0, 15)
                w.ue(ol)
                for _ in range(n):
                    w.u(ol + 1, self.ri(0, (1 << (ol + 1)) - 1))
        if p["header_ext"]:
            n = self.ri(0, 3)
            w.ue(n)
            for _ in range(n):
                w.u(8, self.ri(0, 255))
        w.trailing()                            # byte_alignment()
        hdr = BitWriter()
        hdr.u1(0)
        hdr.u(6, nal_type)
        hdr.u(6, 0)
        hdr.u(3, tid)
        body = bytes(payload)
        if not body or body[-1] == 0:
            body += b"\x80"
        return rbsp_to_nal(hdr.bytes() + w.bytes() + body)

    def pwt(self, w, slice_type, l0, l1):
        s = self.sps
        w.ue(self.ri(0, 7))
        cat = s["chroma_format_idc"] if s["separate_colour_plane_flag"] == 0 else 0
        if cat != 0:
            w.se(self.ri(-2, 2))
        for lx, active in ((0, True), (1, slice_type == 0)):
            if not active:
                continue
            n = (l0 if lx == 0 else l1) + 1
            lw = [self.ri(0, 1) for _ in range(n)]
            for f in lw:
                w.u1(f)
            cw = [0] * n
            if cat != 0:
                cw = [self.ri(0, 1) for _ in range(n)]
                for f in cw:
                    w.u1(f)
            for i in range(n):
                if lw[i]:
                    w.se(self.ri(-20, 20))
                    w.se(self.ri(-20, 20))
                if cw[i]:
                    for _ in range(2):
                        w.se(self.ri(-20, 20))
                        w.se(self.ri(-50, 50))


def annexb(nals, four_byte_every=4):
    out = bytearray()
    for k, n in enumerate(nals):
        out += b"\x00\x00\x00\x01" if k % four_byte_every == 0 else b"\x00\x00\x01"
        out += n
    return bytes(out)


def stream_4k30(seed, n_pictures, slices_per_picture=8, idr_every=60, payload_bytes=(9000, 11000), rich=False, forbidden_every=0):
    """Config 3: synthetic 3840x2160 elementary stream -- VPS+SPS+PPS before each IDR,
    `slices_per_picture` slice segments per picture, P/B pictures using the SPS RPS sets."""
    g = Synth(seed, rich=rich)
    rng = np.random.RandomState(seed + 1)
    nals = []
    count = 0
    for pic in range(n_pictures):
        idr = pic % idr_every == 0
        if idr:
            nals.append(g.vps())
            nals.append(g.sps_nal(3840, 2160, ctb_log2=6))
            nals.append(g.pps_nal(force={"tiles": 0, "lists_mod": 1} if forbidden_every else {"tiles": 0}))
        for sl in range(slices_per_picture):
            n = int(rng.randint(payload_bytes[0], payload_bytes[1]))
            payload = rng.randint(0, 256, size=n).astype(np.uint8).tobytes()
            count += 1
            bad = forbidden_every and not idr and count % forbidden_every == 0
            if bad:      # out of spec on purpose: an IDR coded as a P slice asks for the RPS row the last slice with an own set left behind
                nals.append(g.slice_nal(19, first=(sl == 0), payload=payload, slice_type=1, address=sl * (2040 // slices_per_picture)))
                continue
            nals.append(g.slice_nal(19 if idr else 1, first=(sl == 0), payload=payload,
                                    address=sl * (2040 // slices_per_picture)))
    return annexb(nals), len(nals)
