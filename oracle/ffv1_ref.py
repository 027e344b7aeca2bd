"""oracle/ffv1_ref.py -- an independent, deliberately plain restatement of FFV1 (RFC 9043) and of the Matroska subset around it
(RFC 9559), in Python.  TEST INFRASTRUCTURE ONLY: it checks metric_depth_video_toolbox_amd/csrc_host/mdvt_video.cpp and is never
imported by the product (tests/test_host_cpu.py::test_product_never_imports_the_oracle).  Pure-Python loops: small frames only.

PARITY UNPINNED against FFmpeg itself (none in the image; tests/golden/gen_ffv1_golden.py makes cross-check vectors on a machine
that has one).  What is restated, section by section of the RFC:
  3.8.1   range coder: get_rac / put_rac, byte-wise renormalisation with carry propagation, the 32-state integer binarisation
  3.8.1.3 default_state_transition (generated; its first and last entries are asserted against the RFC's table in the tests)
  3.8.2   Golomb-Rice mode: signed Rice codes with the 12-zero escape, the per-context {drift, error_sum, bias, count} adaptation,
          run mode with the log2_run table
  3.3-3.7 median prediction, the quantised-difference context (3 or 5 inputs, sign folding), the JPEG 2000 RCT, line interleaving
  4.2     configuration record (version 3), 4.3-4.8 frame / slice header / slice footer (24-bit size, error_status, CRC-32 parity);
          versions 0 and 1 carry the same parameters inside every key frame and have a single slice without a footer.
Both directions exist here: the decoder checks the product's encoder; the ENCODER makes streams in every mode the product's decoder
claims (Golomb-Rice, custom state table, versions 0 / 1, inter frames whose contexts carry over, alpha, 5-input contexts), since the
product's own encoder only ever writes one of them.
"""
import struct

import numpy as np

LOG2_RUN = [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24]


def crc32_mpeg(data, crc=0):
    for b in data:
        crc ^= b << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc


def default_state_transition():
    one_state = [0] * 256
    one = 1 << 32
    factor = int(0.05 * (1 << 32))
    max_p = 256 - 8
    last_p8, p = 0, one // 2
    for _ in range(128):
        p8 = (256 * p + one // 2) >> 32
        if p8 <= last_p8:
            p8 = last_p8 + 1
        if last_p8 and last_p8 < 256 and p8 <= max_p:
            one_state[last_p8] = p8
        p += ((one - p) * factor + one // 2) >> 32
        last_p8 = p8
    for i in range(256 - max_p, max_p + 1):
        if one_state[i]:
            continue
        p = (i * one + 128) >> 8
        p += ((one - p) * factor + one // 2) >> 32
        p8 = (256 * p + one // 2) >> 32
        if p8 <= i:
            p8 = i + 1
        if p8 > max_p:
            p8 = max_p
        one_state[i] = p8
    return one_state


DEFAULT_ONE = default_state_transition()


def zero_from_one(one_state):
    z = [0] * 256
    for i in range(1, 256):
        z[i] = (256 - one_state[256 - i]) & 0xFF
    return z


class RangeDecoder:
    def __init__(self, data, one_state=None):
        self.data, self.pos = data, 2
        self.low = (data[0] << 8) | data[1]
        self.range = 0xFF00
        self.end = len(data)
        if self.low >= 0xFF00:
            self.low, self.end = 0xFF00, 2
        self.set_table(one_state or DEFAULT_ONE)

    def set_table(self, one_state):
        self.one, self.zero = one_state, zero_from_one(one_state)

    def get(self, st, i):
        r1 = (self.range * st[i]) >> 8
        self.range -= r1
        if self.low < self.range:
            st[i] = self.zero[st[i]]
            bit = 0
        else:
            self.low -= self.range
            self.range = r1
            st[i] = self.one[st[i]]
            bit = 1
        if self.range < 0x100:
            self.range <<= 8
            self.low <<= 8
            if self.pos < self.end:
                self.low += self.data[self.pos]
                self.pos += 1
        return bit

    def symbol(self, st, base=0, signed=False):
        if self.get(st, base):
            return 0
        e = 0
        while self.get(st, base + 1 + min(e, 9)):
            e += 1
            assert e <= 31
        a = 1
        for i in range(e - 1, -1, -1):
            a = a + a + self.get(st, base + 22 + min(i, 9))
        if signed and self.get(st, base + 11 + min(e, 10)):
            return -a
        return a


class RangeEncoder:
    def __init__(self, one_state=None):
        self.out = bytearray()
        self.low, self.range, self.outstanding_count, self.outstanding_byte = 0, 0xFF00, 0, -1
        self.one = one_state or DEFAULT_ONE
        self.zero = zero_from_one(self.one)

    def set_table(self, one_state):
        self.one, self.zero = one_state, zero_from_one(one_state)

    def _renorm(self):
        while self.range < 0x100:
            if self.outstanding_byte < 0:
                self.outstanding_byte = self.low >> 8
            elif self.low <= 0xFF00:
                self.out.append(self.outstanding_byte)
                self.out.extend(b"\xff" * self.outstanding_count)
                self.outstanding_count = 0
                self.outstanding_byte = self.low >> 8
            elif self.low >= 0x10000:
                self.out.append(self.outstanding_byte + 1)
                self.out.extend(b"\x00" * self.outstanding_count)
                self.outstanding_count = 0
                self.outstanding_byte = (self.low >> 8) & 0xFF
            else:
                self.outstanding_count += 1
            self.low = (self.low & 0xFF) << 8
            self.range <<= 8

    def put(self, st, i, bit):
        r1 = (self.range * st[i]) >> 8
        if not bit:
            self.range -= r1
            st[i] = self.zero[st[i]]
        else:
            self.low += self.range - r1
            self.range = r1
            st[i] = self.one[st[i]]
        self._renorm()

    def symbol(self, st, v, base=0, signed=False):
        if v == 0:
            self.put(st, base, 1)
            return
        a = abs(v)
        e = a.bit_length() - 1
        self.put(st, base, 0)
        for i in range(e):
            self.put(st, base + 1 + min(i, 9), 1)
        self.put(st, base + 1 + min(e, 9), 0)
        for i in range(e - 1, -1, -1):
            self.put(st, base + 22 + min(i, 9), (a >> i) & 1)
        if signed:
            self.put(st, base + 11 + min(e, 10), 1 if v < 0 else 0)

    def terminate(self, sentinel):
        if sentinel:
            self.put([129], 0, 0)
        self.range = 0xFF
        self.low += 0xFF
        self._renorm()
        self.range = 0xFF
        self._renorm()
        return len(self.out)


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, n, v):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


class BitReader:
    def __init__(self, data):
        self.data, self.pos = data, 0

    def get1(self):
        p = self.pos
        self.pos += 1
        return (self.data[p >> 3] >> (7 - (p & 7))) & 1 if (p >> 3) < len(self.data) else 0

    def get(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.get1()
        return v


def fold(v, bits):
    m = 1 << (bits - 1)
    return ((v + m) & ((1 << bits) - 1)) - m


class Vlc:
    def __init__(self):
        self.drift, self.error_sum, self.bias, self.count = 0, 4, 0, 1

    def k(self):
        i, k = self.count, 0
        while i < self.error_sum:
            k += 1
            i += i
        return k

    def update(self, v):
        drift, count = self.drift + v, self.count
        self.error_sum += abs(v)
        if count == 128:
            count >>= 1
            drift >>= 1
            self.error_sum >>= 1
        count += 1
        if drift <= -count:
            self.bias = max(self.bias - 1, -128)
            drift = max(drift + count, -count + 1)
        elif drift > 0:
            self.bias = min(self.bias + 1, 127)
            drift = min(drift - count, 0)
        self.drift, self.count = drift, count


def quant11(i):
    d = i if i < 128 else i - 256
    a = abs(d)
    q = 0 if a == 0 else 1 if a < 2 else 2 if a < 5 else 3 if a < 12 else 4 if a < 32 else 5
    return -q if d < 0 else q


def quant5(i):
    d = i if i < 128 else i - 256
    a = abs(d)
    q = 0 if a == 0 else 1 if a < 3 else 2
    return -q if d < 0 else q


def quant_tables(five):
    """-> (five 256-entry tables, context_count): FFmpeg's 8-bit tables for context model 0 (3 inputs) / 1 (5 inputs)."""
    t = [[0] * 256 for _ in range(5)]
    for i in range(256):
        t[0][i], t[1][i] = quant11(i), 11 * quant11(i)
        if five:
            t[2][i], t[3][i], t[4][i] = 11 * 11 * quant5(i), 5 * 11 * 11 * quant5(i), 5 * 5 * 11 * 11 * quant5(i)
        else:
            t[2][i] = 11 * 11 * quant11(i)
    return t, ((11 * 11 * 5 * 5 * 5 + 1) // 2 if five else (11 * 11 * 11 + 1) // 2)


def write_quant_table(rc, q):
    st = [128] * 32
    last = 0
    for i in range(1, 128):
        if q[i] != q[i - 1]:
            rc.symbol(st, i - last - 1)
            last = i
    rc.symbol(st, 128 - last - 1)


def read_quant_tables(rd):
    tables, scale = [], 1
    for _ in range(5):
        st, q, i, v = [128] * 32, [0] * 256, 0, 0
        while i < 128:
            ln = rd.symbol(st) + 1
            assert ln <= 128 - i
            for _k in range(ln):
                q[i] = scale * v
                i += 1
            v += 1
        for i in range(1, 128):
            q[256 - i] = -q[i]
        q[128] = -q[127]
        tables.append(q)
        scale *= 2 * v - 1
    return tables, (scale + 1) // 2


class Params:
    def __init__(self, version=3, micro=4, coder=1, alpha=0, nh=1, nv=1, five=False, ec=1, intra=1, custom=None):
        self.version, self.micro, self.coder, self.alpha, self.nh, self.nv = version, micro, coder, alpha, nh, nv
        self.quant, self.context_count = quant_tables(five)
        self.ec, self.intra = (ec, intra) if version >= 3 else (0, 0)
        self.one_state = custom if (coder == 2 and custom) else DEFAULT_ONE

    def five(self):
        return self.quant[3][127] != 0 or self.quant[4][127] != 0


def _header_fields(rc_or_rd, p, st, writing):
    """The fields shared by the configuration record and the version 0 / 1 frame header, after `version`."""
    io = rc_or_rd
    if writing:
        io.symbol(st, p.coder)
        if p.coder == 2:
            st2 = [128] * 32 if p.version >= 2 else st
            for i in range(1, 256):
                io.symbol(st2, p.one_state[i] - DEFAULT_ONE[i], signed=True)
        io.symbol(st, 1)
        if p.version > 0:
            io.symbol(st, 8)
        io.put(st, 0, 1)
        io.symbol(st, 0)
        io.symbol(st, 0)
        io.put(st, 0, p.alpha)


def config_record(p):
    rc = RangeEncoder()
    st = [128] * 32
    rc.symbol(st, p.version)
    rc.symbol(st, p.micro)
    _header_fields(rc, p, st, True)
    rc.symbol(st, p.nh - 1)
    rc.symbol(st, p.nv - 1)
    rc.symbol(st, 1)
    for t in p.quant:
        write_quant_table(rc, t)
    rc.put(st, 0, 0)
    rc.symbol(st, p.ec)
    rc.symbol(st, p.intra)
    rc.terminate(False)
    out = bytes(rc.out)
    return out + struct.pack(">I", crc32_mpeg(out))


def parse_config_record(data):
    assert crc32_mpeg(data) == 0, "configuration record CRC"
    rd = RangeDecoder(data[:-4])
    st, st2 = [128] * 32, [128] * 32
    p = Params()
    p.version = rd.symbol(st)
    p.micro = rd.symbol(st)
    p.coder = rd.symbol(st)
    if p.coder == 2:
        p.one_state = [0] + [rd.symbol(st2, signed=True) + DEFAULT_ONE[i] for i in range(1, 256)]
    colorspace, bits = rd.symbol(st), rd.symbol(st)
    chroma, hs, vs = rd.get(st, 0), rd.symbol(st), rd.symbol(st)
    p.alpha = rd.get(st, 0)
    p.nh, p.nv = rd.symbol(st) + 1, rd.symbol(st) + 1
    nq = rd.symbol(st)
    assert (colorspace, bits, chroma, hs, vs, nq) == (1, 8, 1, 0, 0, 1), (colorspace, bits, chroma, hs, vs, nq)
    p.quant, p.context_count = read_quant_tables(rd)
    assert rd.get(st, 0) == 0, "initial states not restated here"
    p.ec, p.intra = rd.symbol(st), rd.symbol(st)
    return p


def median(a, b, c):
    return sorted((a, b, c))[1]


def context_of(q, five, cur, last, x):
    L, LT, T, RT = cur[x - 1], last[x - 1], last[x], last[x + 1]
    ctx = q[0][(L - LT) & 0xFF] + q[1][(LT - T) & 0xFF] + q[2][(T - RT) & 0xFF]
    if five:
        ctx += q[3][(cur[x - 2] - L) & 0xFF] + q[4][(cur[x] - T) & 0xFF]        # cur[x] still holds the line two rows up: TT
    return ctx


class SliceCoder:
    """The state of one slice that survives from frame to frame of an inter-coded stream."""

    def __init__(self, p):
        self.p = p
        self.reset()

    def reset(self):
        n = self.p.context_count
        np_ = 2 + self.p.alpha
        self.states = [[[128] * 32 for _ in range(n)] for _ in range(np_)]
        self.vlc = [[Vlc() for _ in range(n)] for _ in range(np_)]


def _rct_forward(rgb):
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    b, r = b - g, r - g
    g = g + ((b + r) >> 2)
    return [g, b + 256, r + 256]


def encode_slice_body(sc, rc, bw, planes):
    """planes: list of H x w int arrays (Y, Cb + 256, Cr + 256[, A]).  Range coder `rc` or bit writer `bw` (Golomb-Rice)."""
    p = sc.p
    h, w = planes[0].shape
    five = p.five()
    bufs = [[[0] * (w + 6), [0] * (w + 6)] for _ in planes]
    run_index = 0
    for y in range(h):
        for pi, pl in enumerate(planes):
            bufs[pi][0], bufs[pi][1] = bufs[pi][1], bufs[pi][0]
            last, cur = bufs[pi][0], bufs[pi][1]                 # offset 3: index x + 3
            O = 3
            new = [int(v) for v in pl[y]]
            cur[O - 1] = last[O]
            last[O + w] = last[O + w - 1]
            plane_index = 2 if pi == 3 else (pi + 1) // 2
            run_count = run_mode = 0
            for x in range(w):
                # the context reads cur[x] (= TT) before it is overwritten
                L, LT, T, RT = cur[O + x - 1], last[O + x - 1], last[O + x], last[O + x + 1]
                ctx = p.quant[0][(L - LT) & 0xFF] + p.quant[1][(LT - T) & 0xFF] + p.quant[2][(T - RT) & 0xFF]
                if five:
                    ctx += p.quant[3][(cur[O + x - 2] - L) & 0xFF] + p.quant[4][(cur[O + x] - T) & 0xFF]
                diff = new[x] - median(L, T, L + T - LT)
                cur[O + x] = new[x]
                if ctx < 0:
                    ctx, diff = -ctx, -diff
                diff = fold(diff, 9)
                if p.coder:
                    rc.symbol(sc.states[plane_index][ctx], diff, signed=True)
                    continue
                if ctx == 0:
                    run_mode = 1
                if run_mode:
                    if diff:
                        while run_count >= 1 << LOG2_RUN[run_index]:
                            run_count -= 1 << LOG2_RUN[run_index]
                            run_index += 1
                            bw.put(1, 1)
                        bw.put(1 + LOG2_RUN[run_index], run_count)
                        if run_index:
                            run_index -= 1
                        run_count = run_mode = 0
                        if diff > 0:
                            diff -= 1
                    else:
                        run_count += 1
                if run_mode == 0:
                    st = sc.vlc[plane_index][ctx]
                    v = fold(diff - st.bias, 9)
                    k = st.k()
                    code = v ^ ((2 * st.drift + st.count) >> 31)
                    u = 2 * code if code >= 0 else -2 * code - 1
                    if (u >> k) < 12:
                        bw.put((u >> k) + k + 1, (1 << k) + (u & ((1 << k) - 1)))
                    else:
                        bw.put(12 + 9, u - 11)
                    st.update(v)
            if not p.coder and run_mode:
                while run_count >= 1 << LOG2_RUN[run_index]:
                    run_count -= 1 << LOG2_RUN[run_index]
                    run_index += 1
                    bw.put(1, 1)
                if run_count:
                    bw.put(1, 1)


def slice_rect(p, W, H, sx, sy):
    x0, y0 = sx * W // p.nh, sy * H // p.nv
    return x0, y0, (sx + 1) * W // p.nh - x0, (sy + 1) * H // p.nv - y0


class StreamEncoder:
    """frames (H x W x 3 or 4 uint8, RGB[A]) -> FFV1 packets, in any of the modes of Params; key frame every `gop` frames."""

    def __init__(self, p, W, H, gop=1):
        self.p, self.W, self.H, self.gop, self.n = p, W, H, gop, 0
        self.slices = [SliceCoder(p) for _ in range(p.nh * p.nv if p.version >= 3 else 1)]

    def encode(self, frame):
        p = self.p
        key = self.n % self.gop == 0
        self.n += 1
        planes = _rct_forward(frame)
        if p.alpha:
            planes.append(frame[..., 3].astype(np.int32))
        packet = bytearray()
        for i, sc in enumerate(self.slices):
            sx, sy = i % p.nh, i // p.nh
            x0, y0, sw, sh = slice_rect(p, self.W, self.H, sx, sy) if p.version >= 3 else (0, 0, self.W, self.H)
            rc = RangeEncoder()
            if i == 0:
                rc.put([128], 0, 1 if key else 0)
                if key and p.version < 2:
                    st = [128] * 32
                    rc.symbol(st, p.version)
                    _header_fields(rc, p, st, True)
                    for t in p.quant:
                        write_quant_table(rc, t)
            rc.set_table(p.one_state)
            if p.version >= 3:
                st = [128] * 32
                for v in (sx, sy, 0, 0):
                    rc.symbol(st, v)
                for _ in range(2 + p.alpha):
                    rc.symbol(st, 0)
                for v in (3, 0, 0):
                    rc.symbol(st, v)
            if key:
                sc.reset()
            sub = [pl[y0:y0 + sh, x0:x0 + sw] for pl in planes]
            if p.coder:
                encode_slice_body(sc, rc, None, sub)
                rc.terminate(True)
                body = bytes(rc.out)
            else:
                rc.terminate(p.version > 2)
                bw = BitWriter()
                encode_slice_body(sc, None, bw, sub)
                body = bytes(rc.out) + bw.bytes()
            if p.version >= 3:
                body += struct.pack(">I", len(body))[1:]
                if p.ec:
                    body += b"\x00"
                    body += struct.pack(">I", crc32_mpeg(body))
            packet += body
        return bytes(packet)


def decode_frame_v3(packet, p, W, H):
    """One INTRA frame of a version 3 range-coder stream (what the product's encoder writes) -> H x W x 3 uint8 RGB."""
    assert p.version == 3 and p.coder in (1, 2)
    out = np.zeros((H, W, 3), np.uint8)
    n = p.nh * p.nv
    trailer = 3 + (5 if p.ec else 0)
    end, ext = len(packet), [None] * n
    for i in range(n - 1, -1, -1):
        size = int.from_bytes(packet[end - trailer:end - trailer + 3], "big")
        off = end - trailer - size
        assert off >= 0
        if p.ec:
            assert packet[end - 5] == 0, "error_status"
            assert crc32_mpeg(packet[off:end]) == 0, f"slice {i} CRC"
        ext[i] = (off, size)
        end = off
    assert end == 0
    five = p.five()
    for i, (off, size) in enumerate(ext):
        rd = RangeDecoder(packet[off:off + size])
        if i == 0:
            assert rd.get([128], 0) == 1, "key frame"
        rd.set_table(p.one_state)
        st = [128] * 32
        sx, sy, cw, ch = rd.symbol(st), rd.symbol(st), rd.symbol(st) + 1, rd.symbol(st) + 1
        assert (cw, ch) == (1, 1)
        for _ in range(2 + p.alpha):
            assert rd.symbol(st) == 0
        rd.symbol(st), rd.symbol(st), rd.symbol(st)
        x0, y0, sw, sh = slice_rect(p, W, H, sx, sy)
        states = [[[128] * 32 for _ in range(p.context_count)] for _ in range(2 + p.alpha)]
        npl = 3 + p.alpha
        bufs = [[[0] * (sw + 6), [0] * (sw + 6)] for _ in range(npl)]
        O = 3
        for y in range(sh):
            for pi in range(npl):
                bufs[pi][0], bufs[pi][1] = bufs[pi][1], bufs[pi][0]
                last, cur = bufs[pi][0], bufs[pi][1]
                cur[O - 1] = last[O]
                last[O + sw] = last[O + sw - 1]
                pidx = 2 if pi == 3 else (pi + 1) // 2
                for x in range(sw):
                    ctx = context_of(p.quant, five, cur, last, O + x)
                    sign = ctx < 0
                    d = rd.symbol(states[pidx][abs(ctx)], signed=True)
                    if sign:
                        d = -d
                    L, LT, T = cur[O + x - 1], last[O + x - 1], last[O + x]
                    cur[O + x] = (median(L, T, L + T - LT) + d) & 0x1FF
            for x in range(sw):
                g, b, r = bufs[0][1][O + x], bufs[1][1][O + x] - 256, bufs[2][1][O + x] - 256
                g -= (b + r) >> 2
                out[y0 + y, x0 + x] = ((r + g) & 0xFF, g & 0xFF, (b + g) & 0xFF)
    return out


# --------------------------------------------------------------------------------------------------------------- Matroska
def _vint(n, width=None):
    if width is None:
        width = 1
        while n >= (1 << (7 * width)) - 1:
            width += 1
    return ((1 << (7 * width)) | n).to_bytes(width, "big")


def _el(eid, payload):
    return eid.to_bytes((eid.bit_length() + 7) // 8, "big") + _vint(len(payload)) + payload


def _uint(eid, v):
    return _el(eid, v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big"))


def mux_matroska(packets, W, H, fps, codec_private, vfw=False, block_groups=False, frames_per_cluster=3, unknown_cluster_size=False):
    """A minimal Matroska file around FFV1 packets, in the variants the reader claims: CodecID V_FFV1 or V_MS/VFW/FOURCC (a
    BITMAPINFOHEADER in front of the configuration record), SimpleBlocks or BlockGroups, several frames per Cluster, Clusters of
    unknown size."""
    ebml = _el(0x1A45DFA3, _uint(0x4286, 1) + _uint(0x42F7, 1) + _uint(0x42F2, 4) + _uint(0x42F3, 8) + _el(0x4282, b"matroska") +
               _uint(0x4287, 4) + _uint(0x4285, 2))
    dur_ms = len(packets) * 1000.0 / fps
    info = _el(0x1549A966, _uint(0x2AD7B1, 1000000) + _el(0x4489, struct.pack(">d", dur_ms)) + _el(0x4D80, b"ffv1_ref") + _el(0x5741, b"ffv1_ref"))
    if vfw:
        bih = struct.pack("<IiiHH4sIiiII", 40 + len(codec_private), W, H, 1, 24, b"FFV1", W * H * 3, 0, 0, 0, 0)
        cid, priv = b"V_MS/VFW/FOURCC", bih + codec_private
    else:
        cid, priv = b"V_FFV1", codec_private
    entry = _uint(0xD7, 1) + _uint(0x73C5, 1) + _uint(0x83, 1) + _uint(0x9C, 0) + _uint(0x23E383, int(round(1e9 / fps))) + _el(0x86, cid)
    if priv:
        entry += _el(0x63A2, priv)
    entry += _el(0xE0, _uint(0xB0, W) + _uint(0xBA, H))
    tracks = _el(0x1654AE6B, _el(0xAE, entry))
    clusters = b""
    for c0 in range(0, len(packets), frames_per_cluster):
        t0 = int(round(c0 * 1000.0 / fps))
        body = _uint(0xE7, t0)
        for k in range(c0, min(c0 + frames_per_cluster, len(packets))):
            rel = int(round(k * 1000.0 / fps)) - t0
            blk = b"\x81" + struct.pack(">h", rel)
            if block_groups:
                body += _el(0xA0, _el(0xA1, blk + b"\x00" + packets[k]))
            else:
                body += _el(0xA3, blk + b"\x80" + packets[k])
        if unknown_cluster_size:
            clusters += (0x1F43B675).to_bytes(4, "big") + b"\x01" + b"\xff" * 7 + body
        else:
            clusters += _el(0x1F43B675, body)
    seg = info + tracks + clusters
    return ebml + (0x18538067).to_bytes(4, "big") + _vint(len(seg), 8) + seg
