"""TensorFlow tensor-bundle (V2 checkpoint) files written / read without TensorFlow
(`ctc_asr_amd/tf_bundle.py`).  No TensorFlow exists in the build container, so
the format is pinned by the published CRC-32C known answers, hand-assembled table blocks and
round trips."""

import struct

import numpy as np
import pytest

from ctc_asr_amd import hostlib, storage, tf_bundle
from ctc_asr_amd.model import ModelConfig, ParamArena, init_params


def test_crc32c_known_answers():
    # RFC 3720 B.4 / the LevelDB crc32c test
    assert hostlib.crc32c(b'123456789') == 0xE3069283
    assert hostlib.crc32c(bytes(32)) == 0x8A9136AA
    assert hostlib.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert hostlib.crc32c(bytes(range(32))) == 0x46DD794E
    assert hostlib.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert hostlib.crc32c(b'6789', hostlib.crc32c(b'12345')) == 0xE3069283     # chaining
    assert hostlib.crc32c(np.arange(8, dtype=np.uint8)) == hostlib.crc32c(bytes(range(8)))
    crc = hostlib.crc32c(b'foo')
    masked = tf_bundle.masked_crc32c(b'foo')
    assert masked != crc
    rot = (masked - tf_bundle.CRC_MASK_DELTA) & 0xFFFFFFFF
    assert ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF == crc


def test_block_decoding_of_hand_assembled_bytes():
    # LevelDB block: entries (shared, non_shared, value_len, key delta, value), restart array,
    # restart count.  "a"->"1", "ab"->"2" (shares 1 byte), "b"->"" with one restart point.
    block = (bytes([0, 1, 1]) + b'a1' + bytes([1, 1, 1]) + b'b2' + bytes([0, 1, 0]) + b'b' +
             struct.pack('<I', 0) + struct.pack('<I', 1))
    assert list(tf_bundle._block_entries(block)) == [(b'a', b'1'), (b'ab', b'2'), (b'b', b'')]
    builder = tf_bundle._BlockBuilder()
    for key, value in [(b'a', b'1'), (b'ab', b'2'), (b'b', b'')]:
        builder.add(key, value)
    assert builder.finish() == block


def test_snappy_blocks_are_readable():
    # literal 'a', then a 9-byte copy at offset 1 (overlapping its own output)
    assert tf_bundle._snappy_decompress(bytes([10, 0x00, 0x61, 0x15, 0x01])) == b'a' * 10
    # 3-byte literal + 2-byte-offset copy of 4 bytes
    assert tf_bundle._snappy_decompress(bytes([7, 0x08]) + b'xyz' + bytes([0x0E, 3, 0])) \
        == b'xyzxyzx'
    with pytest.raises(ValueError):
        tf_bundle._snappy_decompress(bytes([5, 0x15, 0x01]))


def test_entry_proto_encoding():
    # BundleEntryProto{dtype=DT_FLOAT(1), shape{dim{size=3} dim{size=4}}, offset=48, size=48,
    # crc32c=fixed32}
    raw = tf_bundle._encode_entry(1, (3, 4), 48, 48, 0x01020304)
    assert raw == bytes([0x08, 1, 0x12, 8, 0x12, 2, 0x08, 3, 0x12, 2, 0x08, 4, 0x20, 48,
                         0x28, 48, 0x35, 4, 3, 2, 1])
    entry = tf_bundle._decode_entry(raw)
    assert entry['dtype'] == 1 and entry['shape'] == [3, 4] and entry['offset'] == 48
    assert entry['size'] == 48 and entry['crc32c'] == 0x01020304 and entry['shard_id'] == 0
    scalar = tf_bundle._decode_entry(tf_bundle._encode_entry(9, (), 0, 8, 7))
    assert scalar['shape'] == [] and scalar['dtype'] == 9 and scalar['offset'] == 0


def test_bundle_round_trip_many_variables(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {'global_step': np.array(1234, dtype=np.int64),
               'empty': np.zeros((0, 5), dtype=np.float32),
               'flags/bits': rng.integers(0, 2, size=17).astype(np.bool_)}
    for i in range(300):      # long shared prefixes, several 4 KB table blocks
        tensors['rnn/cudnn_lstm/stack_bidirectional_rnn/cell_{}/bidirectional_rnn/fw/'
                'cudnn_compatible_lstm_cell/kernel'.format(i)] = \
            rng.normal(size=(3, i % 7 + 1)).astype(np.float32)
        tensors['dense/dense_{}/bias'.format(i)] = rng.integers(-5, 5, size=i % 4).astype(np.int32)
    prefix = str(tmp_path / 'model.ckpt-1234')
    tf_bundle.write_bundle(prefix, tensors)
    listing = tf_bundle.list_bundle(prefix)
    assert set(listing) == set(tensors)
    assert listing['global_step'] == (np.dtype(np.int64), ())
    back = tf_bundle.read_bundle(prefix)
    for name, value in tensors.items():
        assert back[name].dtype == value.dtype and back[name].shape == value.shape, name
        assert np.array_equal(back[name], value), name
    some = tf_bundle.read_bundle(prefix, ['global_step'])
    assert list(some) == ['global_step'] and int(some['global_step']) == 1234
    with pytest.raises(KeyError):
        tf_bundle.read_bundle(prefix, ['missing'])
    # table structure: footer magic, sorted keys with the header entry first
    items = tf_bundle._read_table(prefix + '.index')
    assert items[0] == (b'', tf_bundle._HEADER)
    keys = [k for k, _ in items]
    assert keys == sorted(keys) and len(keys) == len(tensors) + 1
    with open(prefix + '.index', 'rb') as handle:
        raw = handle.read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'ckpt')
    tf_bundle.write_bundle(prefix, {'w': np.arange(100, dtype=np.float32)})
    with open(prefix + '.data-00000-of-00001', 'r+b') as handle:
        handle.seek(40)
        handle.write(b'\x7f')
    with pytest.raises(ValueError):
        tf_bundle.read_bundle(prefix)
    assert tf_bundle.read_bundle(prefix, verify=False)['w'].shape == (100,)
    with open(prefix + '.index', 'r+b') as handle:
        handle.seek(3)
        handle.write(b'\x55')
    with pytest.raises(ValueError):
        tf_bundle.read_bundle(prefix)
    with open(prefix + '.index', 'wb') as handle:
        handle.write(b'not a table')
    with pytest.raises(ValueError):
        tf_bundle.read_bundle(prefix)


@pytest.mark.parametrize('kind', ['ds2_lstm', 'ds1_basic'])
def test_model_checkpoint_export_import(tmp_path, kind):
    if kind == 'ds2_lstm':
        cfg = ModelConfig(used_model='ds2', conv_filters=(4, 4, 6), num_units_dense=16,
                          num_layers_rnn=2, num_units_rnn=64, rnn_cell='lstm', cudnn=True)
    else:
        cfg = ModelConfig(used_model='ds1', num_units_dense=16, num_layers_rnn=1,
                          num_units_rnn=64, rnn_cell='rnn_tanh', cudnn=False)
    flat = init_params(cfg, 3)
    rng = np.random.default_rng(1)
    for name in flat:
        if name.endswith('b_hh'):
            continue                      # TensorFlow keeps ONE bias per cell: b_ih + b_hh
        flat[name] = (flat[name] + rng.normal(size=flat[name].shape) * 0.1).astype(np.float32)
    arena = ParamArena(cfg, 'cpu')
    arena.load(flat)
    prefix = storage.export_tf_checkpoint(str(tmp_path), arena, cfg, global_step=77)
    assert prefix.endswith('model.ckpt-77')
    assert tf_bundle.latest_checkpoint(str(tmp_path)) == prefix
    names = tf_bundle.list_bundle(prefix)
    assert names['global_step'] == (np.dtype(np.int64), ())
    assert 'logits/dense/kernel' in names and 'dense4/dense/bias' in names
    other = ParamArena(cfg, 'cpu')
    other.m.fill_(1.0)
    step = storage.import_tf_checkpoint(str(tmp_path), other, cfg)     # via the state file
    assert step == 77 and float(other.m.abs().sum()) == 0.0
    got = other.export()
    for name, value in arena.export().items():
        assert np.array_equal(got[name], value), name
    with pytest.raises(ValueError):
        storage.import_tf_checkpoint(str(tmp_path / 'nothing'), other, cfg)
