"""Minimal NRRD reader / writer with pynrrd's `nrrd.read(path) -> (data, header)` contract (pynrrd is not in the image).

The reference loads every volume with `img, _ = nrrd.read(path)` (PW_NN.py:436, PW_AL.py:749, :911, :944): pynrrd's default
index order is Fortran, i.e. `data.shape == tuple(header['sizes'])` with the FIRST axis fastest in the file.  Supported:
NRRD0001-0005 attached-data files, encodings `raw` and `gzip`, integer and floating types, either endianness.
Detached data files, bzip2, ascii / hex encodings and `line skip` / `byte skip` are rejected loudly."""
import gzip
import io

import numpy as np

_TYPES = {
    'signed char': 'i1', 'int8': 'i1', 'int8_t': 'i1',
    'uchar': 'u1', 'unsigned char': 'u1', 'uint8': 'u1', 'uint8_t': 'u1',
    'short': 'i2', 'short int': 'i2', 'signed short': 'i2', 'signed short int': 'i2', 'int16': 'i2', 'int16_t': 'i2',
    'ushort': 'u2', 'unsigned short': 'u2', 'unsigned short int': 'u2', 'uint16': 'u2', 'uint16_t': 'u2',
    'int': 'i4', 'signed int': 'i4', 'int32': 'i4', 'int32_t': 'i4',
    'uint': 'u4', 'unsigned int': 'u4', 'uint32': 'u4', 'uint32_t': 'u4',
    'longlong': 'i8', 'long long': 'i8', 'long long int': 'i8', 'signed long long': 'i8', 'int64': 'i8', 'int64_t': 'i8',
    'ulonglong': 'u8', 'unsigned long long': 'u8', 'unsigned long long int': 'u8', 'uint64': 'u8', 'uint64_t': 'u8',
    'float': 'f4', 'double': 'f8',
}


def read_header(f):
    magic = f.readline().decode('ascii', 'replace').strip()
    if not magic.startswith('NRRD000'):
        raise ValueError('not a NRRD file (magic %r)' % magic)
    header = {}
    while True:
        line = f.readline()
        if line == b'':
            raise ValueError('NRRD header without the blank line that ends it (detached data is not supported)')
        line = line.decode('ascii', 'replace').rstrip('\r\n')
        if line == '':
            break
        if line.startswith('#'):
            continue
        if ':=' in line:
            k, v = line.split(':=', 1)
            header[k.strip()] = v.strip()
        elif ': ' in line or line.endswith(':'):
            k, v = (line.split(': ', 1) + [''])[:2]
            header[k.strip().lower()] = v.strip()
        else:
            raise ValueError('bad NRRD header line %r' % line)
    return header


def read(path):
    """(data, header): data.shape == sizes, first axis fastest in the file (pynrrd index_order='F')."""
    with open(path, 'rb') as f:
        h = read_header(f)
        for bad in ('data file', 'datafile', 'line skip', 'lineskip', 'byte skip', 'byteskip'):
            if bad in h and h[bad] not in ('0',):
                raise NotImplementedError('NRRD field %r' % bad)
        if h.get('type', '').lower() not in _TYPES:
            raise NotImplementedError('NRRD type %r' % h.get('type'))
        code = _TYPES[h['type'].lower()]
        endian = {'little': '<', 'big': '>'}.get(h.get('endian', 'little').lower())
        if endian is None:
            raise ValueError('NRRD endian %r' % h.get('endian'))
        dt = np.dtype((endian if code[1] != '1' else '|') + code)
        sizes = [int(v) for v in h['sizes'].split()]
        if len(sizes) != int(h.get('dimension', len(sizes))):
            raise ValueError('NRRD sizes / dimension mismatch')
        raw = f.read()
    enc = h.get('encoding', 'raw').lower()
    if enc in ('gzip', 'gz'):
        raw = gzip.GzipFile(fileobj=io.BytesIO(raw)).read()
    elif enc != 'raw':
        raise NotImplementedError('NRRD encoding %r' % enc)
    n = int(np.prod(sizes))
    if len(raw) < n * dt.itemsize:
        raise ValueError('NRRD data shorter than its sizes')
    data = np.frombuffer(raw, dtype=dt, count=n).reshape(sizes, order='F')
    header = dict(h)
    header['sizes'] = np.array(sizes)
    header['dimension'] = len(sizes)
    return np.array(data.astype(dt.newbyteorder('='), copy=False)), header


def write(path, data, encoding='raw'):
    """Attached-data NRRD0004 of `data` (first axis fastest), for fixtures and round trips."""
    data = np.asarray(data)
    inv = {v: k for k, v in (('float', 'f4'), ('double', 'f8'), ('uint8', 'u1'), ('int8', 'i1'), ('int16', 'i2'), ('uint16', 'u2'),
                             ('int32', 'i4'), ('uint32', 'u4'), ('int64', 'i8'), ('uint64', 'u8'))}
    code = data.dtype.str[1:]
    if code not in inv:
        raise NotImplementedError('dtype %s' % data.dtype)
    payload = np.asarray(data, dtype=data.dtype.newbyteorder('<')).tobytes(order='F')
    if encoding == 'gzip':
        buf = io.BytesIO()
        with gzip.GzipFile(fileobj=buf, mode='wb') as g:
            g.write(payload)
        payload = buf.getvalue()
    elif encoding != 'raw':
        raise NotImplementedError(encoding)
    head = 'NRRD0004\ntype: %s\ndimension: %d\nsizes: %s\nendian: little\nencoding: %s\n\n' % (
        inv[code], data.ndim, ' '.join(str(s) for s in data.shape), encoding)
    with open(path, 'wb') as f:
        f.write(head.encode('ascii'))
        f.write(payload)
