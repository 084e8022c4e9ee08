"""Minimal FITS primary-HDU reader/writer (astropy is optional, not required).

Covers what `ReadFits.read_cubefits` (reference utils.py:178-277) and
`ChannelFit.modeltofits` (_channelfit.py:665-718) need: image data of
BITPIX 8/16/32/64/-32/-64 with BSCALE/BZERO, scalar header cards.
"""
from __future__ import annotations

import numpy as np

_BLOCK = 2880
_DTYPES = {8: '>u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}


def _parse_value(text):
    text = text.strip()
    if not text:
        return None
    if text[0] == "'":
        end, out, i = None, [], 1
        while i < len(text):                 # '' inside a string is a quote
            if text[i] == "'":
                if i + 1 < len(text) and text[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                end = i
                break
            out.append(text[i])
            i += 1
        return ''.join(out).rstrip() if end is not None else text[1:].rstrip()
    val = text.split('/', 1)[0].strip()
    if val == 'T':
        return True
    if val == 'F':
        return False
    try:
        return int(val)
    except ValueError:
        pass
    try:
        return float(val.replace('D', 'E').replace('d', 'e'))
    except ValueError:
        return val


def read_primary(path):
    """Return (data, header) of the primary HDU; header is a plain dict."""
    with open(path, 'rb') as f:
        raw = f.read()
    header, pos, done = {}, 0, False
    while not done:
        block = raw[pos:pos + _BLOCK]
        if len(block) < _BLOCK:
            raise ValueError(f'{path}: truncated FITS header')
        pos += _BLOCK
        for i in range(0, _BLOCK, 80):
            card = block[i:i + 80].decode('ascii', errors='replace')
            key = card[:8].strip()
            if key == 'END':
                done = True
                break
            if not key or key in ('COMMENT', 'HISTORY'):
                continue
            if card[8:10] == '= ':
                header[key] = _parse_value(card[10:])
    naxis = int(header.get('NAXIS', 0))
    if naxis == 0:
        return None, header
    shape = tuple(int(header[f'NAXIS{k}']) for k in range(naxis, 0, -1))
    dt = np.dtype(_DTYPES[int(header['BITPIX'])])
    count = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=count, offset=pos).reshape(shape)
    bscale, bzero = header.get('BSCALE', 1), header.get('BZERO', 0)
    if dt.kind in 'iu' and (bscale != 1 or bzero != 0):
        data = data * float(bscale) + float(bzero)
    else:
        data = data.astype(dt.newbyteorder('='))
    return data, header


def _card(key, value):
    if isinstance(value, (bool, np.bool_)):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, (int, np.integer)):
        v = f'{int(value):>20d}'
    elif isinstance(value, (float, np.floating)):
        v = f'{float(value):>20.13E}'
    else:
        s = str(value).replace("'", "''")
        v = f"'{s:<8}'"
    return f'{key:<8}= {v}'[:80].ljust(80)


def write_cube(path, data, header):
    """Write a float64 image HDU; `header` cards are carried over except the
    structural ones, COMMENT and HISTORY (cf. _channelfit.py:707-714)."""
    data = np.asarray(data, dtype='>f8')
    cards = [_card('SIMPLE', True), _card('BITPIX', -64),
             _card('NAXIS', data.ndim)]
    for k in range(data.ndim):
        cards.append(_card(f'NAXIS{k + 1}', data.shape[data.ndim - 1 - k]))
    skip = {'SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND', 'BSCALE', 'BZERO', 'END'}
    for k, v in header.items():
        if k in skip or k.startswith('NAXIS') or 'COMMENT' in k or 'HISTORY' in k:
            continue
        if v is None:
            continue
        cards.append(_card(k[:8], v))
    cards.append('END'.ljust(80))
    head = ''.join(cards).encode('ascii')
    head += b' ' * (-len(head) % _BLOCK)
    body = data.tobytes()
    body += b'\0' * (-len(body) % _BLOCK)
    with open(path, 'wb') as f:
        f.write(head)
        f.write(body)
