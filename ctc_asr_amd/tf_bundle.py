"""Reader / writer for TensorFlow's *tensor bundle* (V2 checkpoint) files, without TensorFlow.

``tf.estimator`` — the reference's trainer (``asr/train.py:31-55``) — keeps its checkpoints as
``<train_dir>/model.ckpt-<global_step>.index`` + ``.data-00000-of-00001`` and a text file
``checkpoint`` naming the latest one.  SURVEY.md 8f-1 asks for import/export of exactly those, so
that identical weights can be fed to both systems.  The format, restated from its public
description (``tensorflow/core/util/tensor_bundle`` and the LevelDB table format it builds on):

* ``.data-*``: the tensors' raw little-endian bytes, back to back.
* ``.index``: an immutable sorted string table.  Key ``""`` -> ``BundleHeaderProto`` (num_shards,
  endianness, version); key = variable name -> ``BundleEntryProto`` (dtype, shape, shard_id,
  offset, size, masked CRC-32C of the bytes).  Table layout: data blocks | meta-index block |
  index block | 48-byte footer (two block handles, padding, magic ``0xdb4775248b80fb57``); a
  block is prefix-compressed entries (varint shared / non-shared / value lengths) + restart
  array + restart count, followed by a 1-byte compression type and the masked CRC-32C of block
  and type.  TensorFlow writes the index uncompressed; snappy blocks are read as well.

UNVERIFIED against TensorFlow itself (none in the build container, no network): checked by
round trips, the CRC-32C known answers and hand-assembled blocks in ``tests/test_tf_bundle.py``.
"""

import os
import struct

import numpy as np

from ctc_asr_amd import hostlib

TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_SIZE = 4096
RESTART_INTERVAL = 16
CRC_MASK_DELTA = 0xa282ead8

# tensorflow DataType enum <-> numpy
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
           9: np.int64, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


def masked_crc32c(data):
    crc = hostlib.crc32c(data)
    return (((crc >> 15) | (crc << 17)) + CRC_MASK_DELTA) & 0xFFFFFFFF


# ------------------------------------------------------------------------------- varints, protos
def _varint(value):
    out = bytearray()
    value &= (1 << 64) - 1
    while value >= 0x80:
        out.append((value & 0x7F) | 0x80)
        value >>= 7
    out.append(value)
    return bytes(out)


def _read_varint(buf, pos):
    result = shift = 0
    while True:
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if byte < 0x80:
            return result, pos
        shift += 7


def _proto_fields(buf):
    """Yield (field number, wire type, value) of a serialized protobuf message."""
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        number, wire = key >> 3, key & 7
        if wire == 0:
            value, pos = _read_varint(buf, pos)
        elif wire == 1:
            value = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            size, pos = _read_varint(buf, pos)
            value = buf[pos:pos + size]
            pos += size
        elif wire == 5:
            value = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type {}'.format(wire))
        yield number, wire, value


def _encode_shape(shape):
    # TensorShapeProto.dim (field 2) = Dim{size (field 1)}; zero is the proto3 default: not written
    return b''.join(b'\x12' + _varint(len(d)) + d
                    for d in ((b'\x08' + _varint(int(size)) if size else b'') for size in shape))


def _encode_entry(dtype_id, shape, offset, size, crc):
    shape_msg = _encode_shape(shape)
    out = b'\x08' + _varint(dtype_id) + b'\x12' + _varint(len(shape_msg)) + shape_msg
    # shard_id 0 is the proto3 default and is not written
    if offset:
        out += b'\x20' + _varint(offset)
    if size:
        out += b'\x28' + _varint(size)
    return out + b'\x35' + struct.pack('<I', crc)


def _decode_entry(buf):
    entry = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None,
             'sliced': False}
    for number, _, value in _proto_fields(buf):
        if number == 1:
            entry['dtype'] = value
        elif number == 2:
            for n2, _, dim in _proto_fields(value):
                if n2 == 2:
                    size = 0
                    for n3, _, v3 in _proto_fields(dim):
                        if n3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    entry['shape'].append(size)
        elif number == 3:
            entry['shard_id'] = value
        elif number == 4:
            entry['offset'] = value
        elif number == 5:
            entry['size'] = value
        elif number == 6:
            entry['crc32c'] = struct.unpack('<I', value)[0]
        elif number == 7:
            entry['sliced'] = True
    return entry


# BundleHeaderProto{num_shards=1, endianness=LITTLE(0, default), version{producer=1}}
_HEADER = b'\x08\x01\x1a\x02\x08\x01'


# ------------------------------------------------------------------------------------ table write
class _BlockBuilder:
    def __init__(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last_key = b''

    def add(self, key, value):
        shared = 0
        if self.count % RESTART_INTERVAL == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            limit = min(len(key), len(self.last_key))
            while shared < limit and key[shared] == self.last_key[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.count += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        tail = b''.join(struct.pack('<I', r) for r in self.restarts)
        return bytes(self.buf) + tail + struct.pack('<I', len(self.restarts))


def _write_block(handle, contents):
    """Append block + trailer (type 0 = uncompressed, masked crc of both); return its handle."""
    offset = handle.tell()
    trailer_type = b'\x00'
    crc = masked_crc32c(contents + trailer_type)
    handle.write(contents + trailer_type + struct.pack('<I', crc))
    return _varint(offset) + _varint(len(contents))


def _write_table(path, items):
    """items: sorted list of (key bytes, value bytes)."""
    with open(path, 'wb') as handle:
        index = _BlockBuilder()
        block = _BlockBuilder()
        for key, value in items:
            block.add(key, value)
            if block.size_estimate() >= BLOCK_SIZE:
                index.add(block.last_key, _write_block(handle, block.finish()))
                block = _BlockBuilder()
        if block.count:
            index.add(block.last_key, _write_block(handle, block.finish()))
        meta_handle = _write_block(handle, _BlockBuilder().finish())
        index_handle = _write_block(handle, index.finish())
        footer = meta_handle + index_handle
        footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
        handle.write(footer)


# ------------------------------------------------------------------------------------- table read
def _snappy_decompress(buf):
    length, pos = _read_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            size = tag >> 2
            if size >= 60:
                extra = size - 59
                size = int.from_bytes(buf[pos:pos + extra], 'little')
                pos += extra
            size += 1
            out += buf[pos:pos + size]
            pos += size
            continue
        if kind == 1:
            size = ((tag >> 2) & 7) + 4
            offset = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            size = (tag >> 2) + 1
            offset = int.from_bytes(buf[pos:pos + 2], 'little')
            pos += 2
        else:
            size = (tag >> 2) + 1
            offset = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if offset == 0 or offset > len(out):
            raise ValueError('corrupt snappy block')
        for _ in range(size):                           # may overlap its own output
            out.append(out[-offset])
    if len(out) != length:
        raise ValueError('corrupt snappy block (length)')
    return bytes(out)


def _read_block(data, handle_offset, handle_size, verify=True):
    contents = data[handle_offset:handle_offset + handle_size]
    kind = data[handle_offset + handle_size]
    if verify:
        stored = struct.unpack('<I', data[handle_offset + handle_size + 1:
                                          handle_offset + handle_size + 5])[0]
        if stored != masked_crc32c(data[handle_offset:handle_offset + handle_size + 1]):
            raise ValueError('tensor bundle index: block checksum mismatch')
    if kind == 1:
        contents = _snappy_decompress(contents)
    elif kind != 0:
        raise ValueError('tensor bundle index: unknown block compression {}'.format(kind))
    return contents


def _block_entries(block):
    num_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        value_len, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + value_len]
        pos += value_len


def _read_table(path, verify=True):
    with open(path, 'rb') as handle:
        data = handle.read()
    if len(data) < 48 or struct.unpack('<Q', data[-8:])[0] != TABLE_MAGIC:
        raise ValueError('{} is not a tensor bundle index (bad magic)'.format(path))
    footer = data[-48:]
    _, pos = _read_varint(footer, 0)            # meta-index handle: offset, size (unused)
    _, pos = _read_varint(footer, pos)
    index_offset, pos = _read_varint(footer, pos)
    index_size, pos = _read_varint(footer, pos)
    items = []
    for _, handle_value in _block_entries(_read_block(data, index_offset, index_size, verify)):
        offset, p2 = _read_varint(handle_value, 0)
        size, _ = _read_varint(handle_value, p2)
        items.extend(_block_entries(_read_block(data, offset, size, verify)))
    return items


# ------------------------------------------------------------------------------------- public API
def write_bundle(prefix, tensors):
    """Write ``{name: ndarray}`` as ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``."""
    items = [(b'', _HEADER)]
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as data:
        for name in sorted(tensors, key=lambda n: n.encode()):
            array = np.asarray(tensors[name])      # (ascontiguousarray would make scalars 1-D)
            if array.dtype.byteorder == '>':
                array = array.astype(array.dtype.newbyteorder('<'))
            dtype_id = _DTYPE_IDS.get(array.dtype)
            if dtype_id is None:
                raise TypeError('{}: dtype {} has no checkpoint encoding here'.format(
                    name, array.dtype))
            raw = array.tobytes()
            data.write(raw)
            items.append((name.encode(), _encode_entry(dtype_id, array.shape, offset, len(raw),
                                                       masked_crc32c(raw))))
            offset += len(raw)
    _write_table(prefix + '.index', items)


def list_bundle(prefix, verify=True):
    """``{name: (numpy dtype, shape)}`` of a checkpoint."""
    out = {}
    for key, value in _read_table(prefix + '.index', verify):
        if key:
            entry = _decode_entry(value)
            out[key.decode()] = (np.dtype(_DTYPES[entry['dtype']]), tuple(entry['shape']))
    return out


def read_bundle(prefix, names=None, verify=True):
    """Read tensors (all, or ``names``) of a checkpoint into ``{name: ndarray}``."""
    items = _read_table(prefix + '.index', verify)
    num_shards = 1
    entries = {}
    for key, value in items:
        if not key:
            for number, _, field in _proto_fields(value):
                if number == 1:
                    num_shards = field
                elif number == 2 and field != 0:
                    raise ValueError('big-endian tensor bundles are not supported')
        else:
            entries[key.decode()] = _decode_entry(value)
    wanted = sorted(entries) if names is None else list(names)
    shards = {}
    out = {}
    for name in wanted:
        if name not in entries:
            raise KeyError('{} not found in checkpoint {}'.format(name, prefix))
        entry = entries[name]
        if entry['sliced']:
            raise ValueError('{}: partitioned variables are not supported'.format(name))
        if entry['dtype'] not in _DTYPES:
            raise TypeError('{}: unsupported checkpoint dtype {}'.format(name, entry['dtype']))
        shard = entry['shard_id']
        if shard not in shards:
            shards[shard] = open('{}.data-{:05d}-of-{:05d}'.format(prefix, shard, num_shards),
                                 'rb')
        shards[shard].seek(entry['offset'])
        raw = shards[shard].read(entry['size'])
        if len(raw) != entry['size']:
            raise ValueError('{}: data file truncated'.format(name))
        if verify and entry['crc32c'] is not None and masked_crc32c(raw) != entry['crc32c']:
            raise ValueError('{}: checksum mismatch'.format(name))
        out[name] = np.frombuffer(raw, dtype=_DTYPES[entry['dtype']]).reshape(entry['shape']) \
            .copy()
    for handle in shards.values():
        handle.close()
    return out


def write_checkpoint_state(directory, latest, all_paths=None):
    """The text-format ``CheckpointState`` file ``tf.train.latest_checkpoint`` reads."""
    lines = ['model_checkpoint_path: "{}"'.format(latest)]
    lines += ['all_model_checkpoint_paths: "{}"'.format(p) for p in (all_paths or [latest])]
    with open(os.path.join(directory, 'checkpoint'), 'w') as handle:
        handle.write('\n'.join(lines) + '\n')


def latest_checkpoint(directory):
    """Prefix of the newest checkpoint named by ``<directory>/checkpoint`` (or None)."""
    state = os.path.join(directory, 'checkpoint')
    if not os.path.isfile(state):
        return None
    with open(state) as handle:
        for line in handle:
            if line.startswith('model_checkpoint_path:'):
                name = line.split(':', 1)[1].strip().strip('"')
                return name if os.path.isabs(name) else os.path.join(directory, name)
    return None
