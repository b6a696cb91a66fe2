"""Kaldi table I/O for the input side of the trainers (the reference vendors Karel Vesely's
kaldi_io.py, data/kaldi_io.py:316-500 -- this is an independent implementation of the same file formats).

Supported: scp / ark tables of binary matrices -- float ('FM '), double ('DM ') and 8-bit compressed ('CM ',
kaldi CompressedMatrix format 1: global min/range header, four uint16 percentiles per column, column-major
uint8 payload) -- plus ASCII matrices, float vectors and `file:offset` / `ark:` rxspecifiers and .gz files.

Two ways to get a matrix:
  * ``read_mat(rx)``      -> float32 ndarray, decoded on the host (what the reference does);
  * ``read_mat_raw(rx)``  -> ``RawMat`` holding the record's payload bytes undecoded; a batch of RawMats is
    shipped as ONE byte blob and decoded + padded (+ 10*log10 + CMVN) on the GPU by
    ``mix_data_loader.collate_kaldi_device`` (re2e_kaldi_decode_pad): 1 byte per element over PCIe instead of 4."""
import gzip
import re
import struct

import numpy as np

U16_TO_UNIT = 1.52590218966964e-05           # 1 / 65535 as kaldi rounds it


class KaldiIOError(Exception):
    pass


class RawMat(object):
    """One matrix record, undecoded.  kind: 'FM', 'DM' or 'CM'; ``payload`` = the record bytes after the type token
    ('FM'/'DM': 10-byte dims + row-major data; 'CM': 16-byte global header + 8*cols column headers + cols*rows bytes)."""
    __slots__ = ('kind', 'rows', 'cols', 'payload')

    def __init__(self, kind, rows, cols, payload):
        self.kind, self.rows, self.cols, self.payload = kind, int(rows), int(cols), payload

    def decode(self):
        return decode_raw(self)


def open_or_fd(rx, mode='rb'):
    """Open ``rx`` (optionally 'ark:' / 'scp:' prefixed, optionally ':offset' suffixed, optionally .gz); file objects pass."""
    if not isinstance(rx, str):
        return rx
    if re.match(r'^(ark|scp)(,\w+)*:', rx):
        rx = rx.split(':', 1)[1]
    offset = None
    m = re.search(r':(\d+)$', rx)
    if m:
        rx, offset = rx[:m.start()], int(m.group(1))
    if rx.endswith('|') or rx.startswith('|'):
        raise KaldiIOError('pipes are not supported: %r' % rx)
    fd = gzip.open(rx, mode) if rx.endswith('.gz') else open(rx, mode)
    if offset is not None:
        fd.seek(offset)
    return fd


def read_key(fd):
    """Next utterance id of an ark stream ('' at end of file)."""
    key = b''
    while True:
        ch = fd.read(1)
        if ch == b'':
            break
        if ch == b' ':
            break
        key += ch
    key = key.decode('latin1').strip()
    if key == '':
        return None
    if not re.match(r'^\S+$', key):
        raise KaldiIOError('bad ark key %r' % key)
    return key


def _read_exact(fd, n):
    buf = fd.read(n)
    if len(buf) != n:
        raise KaldiIOError('truncated record: wanted %d bytes, got %d' % (n, len(buf)))
    return buf


def _read_binary_raw(fd):
    token = _read_exact(fd, 3).decode('latin1')
    if token in ('FM ', 'DM '):
        size = 4 if token == 'FM ' else 8
        dims = _read_exact(fd, 10)
        s1, rows, s2, cols = struct.unpack('<bibi', dims)
        if s1 != 4 or s2 != 4:
            raise KaldiIOError('bad matrix dimension markers')
        data = _read_exact(fd, rows * cols * size)
        return RawMat(token[:2], rows, cols, np.frombuffer(data, np.uint8))
    if token == 'CM ':
        head = _read_exact(fd, 16)
        _, _, rows, cols = struct.unpack('<ffii', head)
        rest = _read_exact(fd, cols * 8 + cols * rows)
        return RawMat('CM', rows, cols, np.frombuffer(head + rest, np.uint8))
    if token.startswith('CM'):
        raise KaldiIOError('compressed matrix formats CM2 / CM3 are not supported (neither are they upstream, kaldi_io.py:414)')
    raise KaldiIOError('unknown matrix header %r' % token)


def decode_raw(raw):
    """RawMat -> float32 (rows, cols) on the host."""
    if raw.kind == 'FM':
        return np.frombuffer(raw.payload.tobytes(), '<f4').reshape(raw.rows, raw.cols).copy()
    if raw.kind == 'DM':
        return np.frombuffer(raw.payload.tobytes(), '<f8').reshape(raw.rows, raw.cols).astype(np.float32)
    buf = raw.payload.tobytes()
    gmin, grange, rows, cols = struct.unpack('<ffii', buf[:16])
    hdr = np.frombuffer(buf[16:16 + 8 * cols], '<u2').reshape(cols, 4)
    perc = (np.float32(gmin) + np.float32(grange) * np.float32(U16_TO_UNIT) * hdr.astype(np.float32)).astype(np.float32)
    data = np.frombuffer(buf[16 + 8 * cols:], np.uint8).reshape(cols, rows)
    p0, p25, p75, p100 = (perc[:, i:i + 1] for i in range(4))
    v = data.astype(np.float32)
    out = np.where(data <= 64, p0 + (p25 - p0) / np.float32(64.) * v,
                   np.where(data <= 192, p25 + (p75 - p25) / np.float32(128.) * (v - 64), p75 + (p100 - p75) / np.float32(63.) * (v - 192)))
    return np.ascontiguousarray(out.T.astype(np.float32))


def _read_ascii(fd):
    rows = []
    while True:
        line = fd.readline().decode('latin1')
        if line == '':
            raise KaldiIOError('end of file inside an ASCII matrix')
        toks = line.strip().split()
        if not toks:
            continue
        if toks[-1] == ']':
            if len(toks) > 1:
                rows.append(np.array(toks[:-1], np.float32))
            return np.vstack(rows) if rows else np.zeros((0, 0), np.float32)
        rows.append(np.array(toks, np.float32))


def read_mat_raw(rx):
    """One matrix record, undecoded (binary records only)."""
    fd = open_or_fd(rx)
    try:
        if _read_exact(fd, 2) != b'\x00B':
            raise KaldiIOError('read_mat_raw needs a binary record')
        return _read_binary_raw(fd)
    finally:
        if fd is not rx:
            fd.close()


def read_mat(rx):
    """One matrix as float32 (binary FM / DM / CM or ASCII)."""
    fd = open_or_fd(rx)
    try:
        flag = _read_exact(fd, 2)
        if flag == b'\x00B':
            return decode_raw(_read_binary_raw(fd))
        if flag != b' [':
            raise KaldiIOError('neither a binary nor an ASCII matrix')
        return _read_ascii(fd)
    finally:
        if fd is not rx:
            fd.close()


def read_mat_ark(rx, raw=False):
    """Generator of (key, matrix) -- or (key, RawMat) with ``raw`` -- over an ark file / stream."""
    fd = open_or_fd(rx)
    try:
        key = read_key(fd)
        while key:
            flag = _read_exact(fd, 2)
            if flag == b'\x00B':
                rec = _read_binary_raw(fd)
                yield key, (rec if raw else decode_raw(rec))
            elif flag == b' [' and not raw:
                yield key, _read_ascii(fd)
            else:
                raise KaldiIOError('bad record for key %s' % key)
            key = read_key(fd)
    finally:
        if fd is not rx:
            fd.close()


def read_mat_scp(rx, raw=False):
    """Generator of (key, matrix | RawMat) over an scp file ('key path[:offset]' per line)."""
    fd = open_or_fd(rx)
    try:
        for line in fd:
            line = line.decode('latin1').strip()
            if not line:
                continue
            key, path = line.split(None, 1)
            yield key, (read_mat_raw(path) if raw else read_mat(path))
    finally:
        if fd is not rx:
            fd.close()


def read_vec_flt(rx):
    fd = open_or_fd(rx)
    try:
        flag = _read_exact(fd, 2)
        if flag == b'\x00B':
            token = _read_exact(fd, 3).decode('latin1')
            size = {'FV ': 4, 'DV ': 8}.get(token)
            if size is None:
                raise KaldiIOError('unknown vector header %r' % token)
            marker, n = struct.unpack('<bi', _read_exact(fd, 5))
            v = np.frombuffer(_read_exact(fd, n * size), '<f4' if size == 4 else '<f8')
            return v.astype(np.float32)
        rest = (flag + fd.readline()).decode('latin1').strip().split()
        return np.array([t for t in rest if t not in ('[', ']')], np.float32)
    finally:
        if fd is not rx:
            fd.close()


def _write_header(fd, key):
    if key:
        fd.write((key + ' ').encode('latin1'))
    fd.write(b'\x00B')


def write_mat(file_or_fd, m, key=''):
    """Binary float32 / float64 matrix record (kaldi_io.py:458-500 format)."""
    fd = open_or_fd(file_or_fd, 'wb')
    try:
        m = np.ascontiguousarray(m)
        if m.dtype == np.float32:
            token = b'FM '
        elif m.dtype == np.float64:
            token = b'DM '
        else:
            raise KaldiIOError('write_mat takes float32 or float64, got %s' % m.dtype)
        _write_header(fd, key)
        fd.write(token + struct.pack('<bibi', 4, m.shape[0], 4, m.shape[1]) + m.tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


def compress_mat(m):
    """float matrix -> bytes of a 'CM ' record body (kaldi CompressedMatrix, format 1): the inverse of the decoder above."""
    m = np.asarray(m, np.float32)
    rows, cols = m.shape
    gmin, gmax = float(m.min()), float(m.max())
    grange = max(gmax - gmin, 1e-30)
    srt = np.sort(m, axis=0)
    q = lambda f: srt[min(rows - 1, int(f * (rows - 1) + 0.5))]
    perc = np.stack([srt[0], q(0.25), q(0.75), srt[-1]], 1)                       # (cols, 4)
    u16 = np.clip(np.rint((perc - gmin) / grange * 65535.0), 0, 65535).astype(np.uint16)
    for i in range(1, 4):                                                         # strictly usable segments
        u16[:, i] = np.maximum(u16[:, i], u16[:, i - 1] + (1 if i < 3 else 1))
    u16 = np.minimum(u16, 65535).astype(np.uint16)
    p = (np.float32(gmin) + np.float32(grange) * np.float32(U16_TO_UNIT) * u16.astype(np.float32)).astype(np.float32)
    p0, p25, p75, p100 = (p[:, i][None, :] for i in range(4))
    eps = np.float32(1e-30)
    lo = np.rint((m - p0) / np.maximum(p25 - p0, eps) * 64.0)
    mid = 64 + np.rint((m - p25) / np.maximum(p75 - p25, eps) * 128.0)
    hi = 192 + np.rint((m - p75) / np.maximum(p100 - p75, eps) * 63.0)
    code = np.where(m < p25, np.clip(lo, 0, 64), np.where(m < p75, np.clip(mid, 65, 192), np.clip(hi, 193, 255))).astype(np.uint8)
    return struct.pack('<ffii', gmin, grange, rows, cols) + u16.astype('<u2').tobytes() + np.ascontiguousarray(code.T).tobytes()


def write_mat_compressed(file_or_fd, m, key=''):
    """8-bit compressed ('CM ') matrix record."""
    fd = open_or_fd(file_or_fd, 'wb')
    try:
        _write_header(fd, key)
        fd.write(b'CM ' + compress_mat(m))
    finally:
        if fd is not file_or_fd:
            fd.close()
